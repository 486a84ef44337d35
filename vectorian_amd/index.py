"""Index family: `Index.find` and the MI355X brute-force index.

Mirrors vectorian/index.py: Query (:25), Match / PyMatch (:192, :382), Index (:434) with
`make_query` (:461-477) and `find` (:479-501), and BruteForceIndex (:509-560) -- whose `_find`
(one core.Document.find per document on a thread pool + ResultSet.extend) is what
HipBruteForceIndex replaces with one vk_query call against the corpus resident in HBM.
"""

import collections
import logging
import time

import numpy as np

from vectorian_amd import core
from vectorian_amd.alignment import GapCost
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim

Region = collections.namedtuple("Region", ["s", "match", "gap_penalty"])
TokenMatch = collections.namedtuple("TokenMatch", ["pos_s", "edges"])
TokenMatchEdge = collections.namedtuple("TokenMatchEdge", ["t", "flow", "distance", "metric"])
TokenMatchT = collections.namedtuple("TokenMatchT", ["text", "index", "pos"])
PartitionData = collections.namedtuple("PartitionData", ["level", "window_size", "window_step"])


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


def _vocab_entries(ids, n):
	"""joint-vocabulary view of one document of a slice (BOWBuilder, vectorian/core/cpp/alignment/bow.h:204-275):
	entries in ascending token id with their positions; without ids every position is an entry of its own
	(UniqueTokensBOWBuilder, :281-333)"""
	if ids is None:
		return [[i] for i in range(n)]
	groups = {}
	for i, t in enumerate(ids):
		groups.setdefault(int(t), []).append(i)
	return [groups[t] for t in sorted(groups)]


def rwmd_sparse_flow(S, ids_s, ids_t, injective, symmetric, normalize_bow):
	"""SparseFlow of the relaxed WMD (RelaxedSolver, vectorian/core/cpp/alignment/wmd.h:287-416): the edges of the
	tighter direction, expanded to positions.  S[i][j]: similarity of slice token i and query token j."""
	len_s, len_t = S.shape
	docs = [_vocab_entries(ids_s, len_s), _vocab_entries(ids_t, len_t)]       # 0 = s, 1 = t
	lens = (len_s, len_t)
	bow = [[float(len(e)) / (lens[c] if normalize_bow else 1.0) for e in docs[c]] for c in (0, 1)]
	def dist(es, et):   # first positions stand for the entry (wmd.h:107-135)
		return max(1.0 - float(S[es[0], et[0]]), 0.0)
	cost, tighter, edges_by_dir = 0.0, 0, [[], []]
	for c, (d1, d2) in enumerate(((1, 0), (0, 1))):          # c = 0 moves t -> s first (wmd.h:303-306)
		acc = 0.0
		for a, src in enumerate(docs[d1]):
			ds = [dist(tgt, src) if d1 == 1 else dist(src, tgt) for tgt in docs[d2]]
			if injective:
				b = int(np.argmin(ds)) if ds else -1
				d = ds[b] if b >= 0 else 1.0
				acc += bow[d1][a] * d
				edges_by_dir[c].append((a, b, bow[d1][a], d))
			else:
				remaining = bow[d1][a]
				for b in sorted(range(len(ds)), key=lambda x: (ds[x], docs[d2][x][0])):
					if remaining <= bow[d2][b]:
						acc += remaining * ds[b]
						edges_by_dir[c].append((a, b, remaining, ds[b]))
						break
					remaining -= bow[d2][b]
					acc += bow[d2][b] * ds[b]
					edges_by_dir[c].append((a, b, bow[d2][b], ds[b]))
				if remaining > 0.0:
					acc += remaining   # wmd.h:373-375 as written
		if not normalize_bow:
			acc /= float(lens[d1])
		if not symmetric:
			tighter, cost = 0, acc
			break
		if acc > cost:
			tighter, cost = c, acc
	source, target, flow, distv = [], [], [], []
	d1 = 1 if tighter == 0 else 0
	for a, b, f, d in edges_by_dir[tighter]:
		if b < 0:
			continue
		s_entry = docs[0][b] if tighter == 0 else docs[0][a]
		t_entry = docs[1][a] if tighter == 0 else docs[1][b]
		nf = f / (1.0 if normalize_bow else bow[d1][a])
		for t in t_entry:
			for s_ in s_entry:
				source.append(t); target.append(s_); flow.append(nf); distv.append(d)
	return {"type": "sparse", "source": np.array(source, dtype=np.int16), "target": np.array(target, dtype=np.int16),
		"flow": np.array(flow, dtype=np.float32), "dist": np.array(distv, dtype=np.float32)}


def dense_flow(S, G, ids_s, ids_t, mass_t):
	"""DenseFlow of an exact transport (FullSolver, wmd.h:228-248; WRD::compute, wrd.h:120-135): flow[t][s] = plan of
	the vocabulary pair / mass of the query entry, dist[t][s] = their distance.  G[j][i]: plan between positions."""
	len_s, len_t = S.shape
	es, et = _vocab_entries(ids_s, len_s), _vocab_entries(ids_t, len_t)
	flow = np.zeros((len_t, len_s), dtype=np.float32)
	distv = np.ones((len_t, len_s), dtype=np.float32)
	for a, te in enumerate(et):
		m = float(sum(mass_t[t] for t in te))
		for b, se in enumerate(es):
			g = float(sum(G[t, s_] for t in te for s_ in se))
			d = max(1.0 - float(S[se[0], te[0]]), 0.0)
			for t in te:
				for s_ in se:
					flow[t, s_] = g / m if m > 0 else 0.0
					distv[t, s_] = d
	return {"type": "dense", "flow": flow, "dist": distv}


def _rows_room(top, i):
	"""slice tokens the similarity rows of winner i have room for: the rows per winner the backend returned; on a sharded index what
	the rank that scored the winner returned (shards.rows_allreduce)"""
	room = getattr(top, "rows_room", None)
	return top.sim_rows.shape[1] if room is None else int(room[i])


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


class AllSlices:
	"""debug = AllSlices(hook): call the debug hook for EVERY slice the search scores, as the reference does (call_debug_hook,
	vectorian/core/cpp/metric/alignment.h:145-173; match/matcher_impl.h:137-170), instead of for the k winners only.  The fused
	scoring kernel keeps no per-slice matrices, so the slices are restated `chunk` at a time after the search: exact, opt-in, slow."""

	all_slices = True

	def __init__(self, hook, chunk=512):
		if not callable(hook):
			raise TypeError("debug must be callable: hook(name, data)")
		self.hook, self.chunk = hook, int(chunk)

	def __call__(self, name, data):
		return self.hook(name, data)


class Index:
	def __init__(self, partition, sim):
		self._partition = partition
		self._sim = sim
		if not partition.contiguous:
			logging.warning("the used partition is non-contiguous, you will miss parts of the content.")

	@property
	def partition(self):
		return self._partition

	@property
	def session(self):
		return self._partition.session

	@property
	def sim(self):
		return self._sim

	def make_query(self, text, n=10, min_score=0.0, debug=None, options: dict = dict()):
		# vectorian/index.py:461-477
		options = options.copy()
		options["max_matches"] = n
		options["min_score"] = min_score
		if debug is not None:
			options["debug"] = debug
		options["partition"] = self._partition.to_args()
		if self._sim is not None:
			sim_args = self._sim.to_args(self)
			if sim_args:
				options["metric"] = sim_args
		return Query(self, self._partition.session.vocab, text, options)

	def find(self, text, n=10, min_score=0.0, debug=None, disable_progress=False,
			run_task=None, make_result=None, options: dict = dict()):
		# vectorian/index.py:479-501
		start_time = time.time()
		query = self.make_query(text, n=n, min_score=min_score, debug=debug, options=options)
		session = self._partition.session
		if make_result is None:
			make_result = session.make_result
		if run_task is None:
			run_task = lambda task: session.on_progress(task, disable_progress=disable_progress)
		matches = run_task(lambda progress: self._find(query, progress=progress))
		return make_result(self, matches, duration=time.time() - start_time)

	def _find(self, query, progress=None):
		raise NotImplementedError()


_QUERY_OPTION_WHITELIST = {
	# Query::initialize (vectorian/core/cpp/query.cpp:45-55)
	"metric", "pos_filter", "tag_filter", "submatch_weight", "bidirectional",
	"max_matches", "min_score", "partition", "debug"}


def _split_gap(gap):
	"""gap_cost is one GapCost or {'s': .., 't': ..} (metric/alignment.h:365-370; vectorian/alignment.py:78-83)"""
	if isinstance(gap, dict):
		from vectorian_amd.alignment import ConstantGapCost
		return gap.get("s", ConstantGapCost(0)), gap.get("t", ConstantGapCost(0))
	return gap, gap


class HipBruteForceIndex(Index):
	"""Brute-force search over every slice of the session, on one MI355X.

	Constructor signature of BruteForceIndex (vectorian/index.py:509-524):
	(partition, sim, *, nlp, saliency=None); `device` selects the GPU.  The whole corpus is
	uploaded once (token tiles in HBM); every `find` is one vk_query."""

	def __init__(self, partition, sim, *, nlp=None, saliency=None, device=0, corpus_factory=None, shard=None, group=None, precision="bf16"):
		"""shard = (rank, world): this process keeps only its contiguous range of the slices in HBM (one process per
		GPU, every process holds the same Session); `find` then merges the ranks' result sets with one all-gather
		of k records over `group` (torch.distributed; backend "nccl" = RCCL over xGMI) and returns the same matches
		on every rank (the reference merges per-document ResultSets, result_set.h:70-93).
		precision = "bf16": unit vectors rounded to bf16 in HBM (the fast path); "f32": the reference's own
		precision (CosineSim is an fp32 sgemm, vectorian/sim/vector.py:66-78), twice the bytes per token."""
		super().__init__(partition, sim)
		self._nlp = nlp
		self._shard, self._group = shard, group
		if not isinstance(sim, OptimizedSpanSim):
			raise TypeError(f"{type(sim).__name__}: the HIP index implements OptimizedSpanSim")
		token_sim = sim.token_sim
		if not isinstance(token_sim, EmbeddingTokenSim):
			raise NotImplementedError("token similarity modifiers are outside the HIP path (SURVEY 2.1)")
		if not isinstance(token_sim.similarity, CosineSim):
			raise NotImplementedError(f"{token_sim.similarity.name}: the HIP path computes CosineSim")
		self._embedding = token_sim.embedding
		self._metric_name = token_sim.to_args(self)["name"]
		session = self.session
		level, size = partition.level, partition.window_size
		# slices: Spans::iterate (vectorian/core/cpp/document.h:147-169) over every document
		starts, ends = [], []
		self._slice_doc, self._slice_id, self._slice_token_at = [], [], []
		base = 0
		for di, doc in enumerate(session.documents):
			st, en = doc.spans[level]["start"], doc.spans[level]["end"]
			n = len(st)
			if n and (st[0] != 0 or (n > 1 and (st[1:] != en[:-1]).any()) or en[-1] != doc.n_tokens):
				raise ValueError("spans must be contiguous and cover the document (document.h:151-168)")
			for sid in range(0, n, partition.window_step):
				j = min(sid + size - 1, n - 1)
				starts.append(base + int(st[sid]))
				ends.append(base + int(en[j]))
				self._slice_doc.append(di)
				self._slice_id.append(sid)
				self._slice_token_at.append(int(st[sid]))
			base += doc.n_tokens
		self._slice_start = np.array(starts, dtype=np.int64)
		self._slice_end = np.array(ends, dtype=np.int64)
		# contiguous partition (window_step == window_size): CSR offsets; otherwise general slices
		contiguous = len(starts) > 0 and starts[0] == 0 and ends[-1] == base and (self._slice_start[1:] == self._slice_end[:-1]).all()
		self._sent_off = np.concatenate(([0], self._slice_end)).astype(np.int64) if contiguous else None
		self._slice_doc = np.array(self._slice_doc, dtype=np.int64)
		n_tokens, n_slices = base, len(self._slice_doc)

		self._boost = None
		if saliency is not None:
			# Booster (vectorian/saliency.py:141-154 -> core.Booster): one multiplicative weight per slice
			self._boost = np.ascontiguousarray(saliency, dtype=np.float32)
			if self._boost.shape != (n_slices,):
				raise ValueError("saliency must hold one float per slice")

		# this process's part of the corpus: slices [sa, sb), tokens [t0, t1)
		self._slice_off, t0, t1 = 0, 0, n_tokens
		dev_start, dev_end, dev_off, dev_boost = self._slice_start, self._slice_end, self._sent_off, self._boost
		if shard is not None:
			from vectorian_amd import shards
			rank, world = shard
			sa, sb = shards.shard_ranges_by_tokens(self._slice_end - self._slice_start, world)[rank]   # equal token counts per rank
			self._slice_off = sa
			t0 = int(self._slice_start[sa]) if sb > sa else 0
			t1 = int(self._slice_end[sb - 1]) if sb > sa else 0
			dev_start, dev_end = self._slice_start[sa:sb] - t0, self._slice_end[sa:sb] - t0
			dev_off = np.concatenate(([0], dev_end)).astype(np.int64) if self._sent_off is not None else None
			dev_boost = None if self._boost is None else np.ascontiguousarray(self._boost[sa:sb])
			n_slices_dev, n_tokens_dev = sb - sa, t1 - t0
		else:
			n_slices_dev, n_tokens_dev = n_slices, n_tokens
		self._dev_boost = dev_boost
		self._n_local, self._n_local_tokens, self._n_tokens = n_slices_dev, n_tokens_dev, n_tokens
		self._xdev = None   # where exchanged records live: the process group's own device (RCCL: this GPU; gloo: the host)

		make = corpus_factory or core.Corpus
		emb = self._embedding
		self._token_ids = None   # static layout: token id per corpus token (joint vocabularies of the transport flows)
		if emb.is_static:
			vocab_vectors = emb.encode_tokens(session.vocab.tokens)
			# magnitudes are kept for WordRotatorsDistance (metric/static.cpp:69-73, 80-120); rows are normalised on upload
			self._corpus = make(layout=core.VK_LAYOUT_STATIC, d=emb.dimension, n_tokens=n_tokens_dev, n_sentences=n_slices_dev,
				vocab_size=max(1, session.vocab.size), keep_magnitudes=True, device=device, precision=precision)
			E = vocab_vectors.unmodified if session.vocab.size else np.zeros((1, emb.dimension), np.float32)
			self._corpus.append_vectors(E, normalize=True)
			ids = np.concatenate([session.doc_token_ids(i) for i in range(len(session.documents))]) if n_tokens else np.zeros(0, np.int32)
			self._corpus.set_token_ids(np.ascontiguousarray(ids[t0:t1]))
			self._token_ids = np.asarray(ids, dtype=np.int32)
		elif emb.is_contextual:
			from vectorian_amd.embedding import Vectors
			self._corpus = make(layout=core.VK_LAYOUT_CONTEXTUAL, d=emb.dimension, n_tokens=n_tokens_dev, n_sentences=n_slices_dev,
				keep_magnitudes=True, device=device, precision=precision)
			doc_base = 0
			for doc in session.documents:
				a, b = max(t0, doc_base), min(t1, doc_base + doc.n_tokens)   # this document's tokens inside the shard
				if b > a:
					self._corpus.append_vectors(Vectors(doc.contextual_vectors(emb.name)).unmodified[a - doc_base:b - doc_base], normalize=True)
				doc_base += doc.n_tokens
		else:
			raise TypeError(emb)
		self._has_pos = all(doc.pos is not None for doc in session.documents) and len(session.documents) > 0
		self._has_tags = all(doc.tags is not None for doc in session.documents) and len(session.documents) > 0
		self._pos_codes = self._tag_codes = None   # per corpus token, host copies (token filters: index maps of the winners)
		if self._has_pos:
			self._pos_codes = np.array([session.pos_code(x) for doc in session.documents for x in doc.pos], dtype=np.int8)
			if n_tokens_dev:
				self._corpus.set_token_pos(self._pos_codes[t0:t1])
		if self._has_tags:
			self._tag_codes = np.array([session.tag_code(x) for doc in session.documents for x in doc.tags], dtype=np.int8)
			if n_tokens_dev:
				self._corpus.set_token_tags(self._tag_codes[t0:t1])
		self._filtered = collections.OrderedDict()   # (pos_mask, tag_mask) -> filtered corpus; the last two filters stay resident
		self._views = []   # further handles on the resident corpus (find_many)
		if dev_off is not None:
			self._corpus.set_sentences(dev_off)
		else:
			self._corpus.set_slices(dev_start, dev_end)
		self._corpus.finalize()
		self._max_slice_len = int((np.asarray(dev_end) - np.asarray(dev_start)).max()) if len(dev_start) else 0   # of the resident part

	@property
	def metric_name(self):
		return self._metric_name

	@property
	def n_slices(self):
		return len(self._slice_doc)

	@property
	def corpus(self):
		return self._corpus

	def _backend_args(self, options):
		"""option dicts of the reference -> vk_query arguments (Query::initialize,
		vectorian/core/cpp/query.cpp:32-154; create_alignment_matcher, metric/alignment.h:780-920)"""
		for k in options:
			if k not in _QUERY_OPTION_WHITELIST:
				raise RuntimeError(f"illegal option {k}")   # query.cpp:60-63
		metric = options.get("metric")
		if not isinstance(metric, dict) or metric.get("metric") not in ("alignment-isolated", "alignment-tag-weighted"):
			raise RuntimeError(f"unknown sentence metric type {metric.get('metric') if isinstance(metric, dict) else metric}")  # instantiate.cpp:191-196
		alignment = metric["alignment"]
		args = dict(
			max_matches=int(options.get("max_matches", 100)),      # query.cpp:87-89
			min_score=float(options.get("min_score", 0.2)),        # query.cpp:91-93
			submatch_weight=float(options.get("submatch_weight", 0.0)),
			bidirectional=bool(options.get("bidirectional", False)))
		algorithm = alignment.get("algorithm")
		if algorithm == "pyalign":
			o = alignment.get("options", {})
			gs, gt = _split_gap(o.get("gap_cost"))
			for g in (gs, gt):
				if not isinstance(g, GapCost):
					raise TypeError(f"gap cost {g!r} is not a GapCost")
			args.update(algorithm=core.VK_ALG_ALIGN, locality=int(o.get("locality", core.Locality.LOCAL)), gap_s=gs, gap_t=gt)
			gaps = (gs, gt)
		elif algorithm == "word-movers-distance":
			args.update(algorithm=core.VK_ALG_RWMD, wmd_full=not alignment.get("relaxed", True),
				rwmd=(alignment["injective"], alignment["symmetric"], alignment["normalize_bow"]))
			gaps = (lambda k: 0.0, lambda k: 0.0)   # gap_cost_s/t of WordMoversDistance return 0 (metric/alignment.h:632-638)
		elif algorithm == "word-rotators-distance":
			args.update(algorithm=core.VK_ALG_WRD, wrd_normalize=alignment.get("normalize_magnitudes", True))
			gaps = (lambda k: 0.0, lambda k: 0.0)
		else:
			raise RuntimeError(f"unknown alignment algorithm {algorithm}")   # metric/alignment.h:914-919
		if metric["metric"] == "alignment-tag-weighted":   # any matcher: TagWeightedSlice wraps the slice (match/instantiate.cpp:173-189)
			args["tag_weighted"] = dict(
				tag_weights=metric["tag_weights"],
				pos_mismatch_penalty=float(metric.get("pos_mismatch_penalty", 0)),
				similarity_threshold=float(metric.get("similarity_threshold", 0)))
		return args, gaps

	def find_many(self, texts, n=10, min_score=0.0, options: dict = dict(), in_flight=3, abort=None, batch=None, progress=None):
		"""Several queries, `in_flight` of them at a time on as many handles of the resident corpus (vk_corpus_view:
		shared arrays, own stream and workspaces) from as many host threads: the selection, traceback and host part of
		one query run beside the scoring kernel of the next (bench.py measures the path this way).  Returns one Result
		per text, in order; each equals what `find` returns.  Not part of the reference's Index (which has one
		ThreadPool task per document inside a single find, vectorian/index.py:544-558).
		batch: None (default) -- queries that can share a call do: over contextual embeddings, alignments go to the backend
		several per call (vk_query_batch: every token tile is read once per pair of queries) and relaxed-WMD queries up to 256
		per call (one MFMA-bound GEMM pass over the corpus per call, BASELINE config 4); False -- never; True -- or raise.
		A sharded index (shard = (rank, world)) takes the same paths: every rank scores its shard (batched calls included), the
		result sets of a chunk of queries travel in ONE all-gather, issued in query order from the calling thread while the
		worker threads score the next queries (the exchange bench.py times), and every rank returns the merged matches.
		progress: called with done / total after every query (every chunk of a batched call) is complete, as the reference
		reports done / total tokens after every document (vectorian/index.py:541-558)."""
		session = self.session
		queries = [self.make_query(t, n=n, min_score=min_score, options=options) for t in texts]
		if abort is not None:
			# one flag for the whole call (an int32 array of one element, as Query._abort): raising it from another thread drops
			# the queries that have not started yet -- they return no matches (Query::abort polled between the queries)
			if not (isinstance(abort, np.ndarray) and abort.dtype == np.int32 and abort.size >= 1):
				raise TypeError("abort must be an int32 numpy array")
			for q in queries:
				q._abort = abort
		batches = self._batch_plan(queries, options) if batch is not False else None
		if batch is True and batches is None:
			raise RuntimeError("find_many(batch=True): these queries cannot share a call (static embeddings, filters, tag weights, "
				"a debug hook, exact transport, or queries that differ in their options)")
		start = time.time()
		if batches is not None:
			results = self._find_batches(queries, batches, in_flight, progress)
		else:
			results = self._find_pipelined(queries, in_flight, progress)
		duration = (time.time() - start) / max(1, len(queries))
		return [session.make_result(self, m, duration=duration) for m in results]

	def _handles(self, n):
		"""the resident corpus and n - 1 further handles on it (own stream and workspaces each)"""
		if not hasattr(self._corpus, "view"):
			return [self._corpus]
		while len(self._views) < n - 1:
			self._views.append(self._corpus.view())
		return [self._corpus] + self._views[:max(0, n - 1)]

	def _in_order(self, n_items, n_lanes, work, finish, group=1):
		"""work(item, lane) on n_lanes worker threads (lane l serves items l, l + n_lanes, .. in order: a handle is used by one
		thread); finish(first, outputs) on the CALLING thread for runs of `group` consecutive items, in item order -- the
		collectives of a sharded index must be issued in the same order on every rank, whatever the threads' timing."""
		from concurrent.futures import Future, ThreadPoolExecutor
		futs = [Future() for _ in range(n_items)]

		def lane(l):
			for i in range(l, n_items, n_lanes):
				try:
					futs[i].set_result(work(i, l))
				except BaseException as e:   # surfaces on the calling thread, in order
					futs[i].set_exception(e)
		with ThreadPoolExecutor(max_workers=max(1, n_lanes)) as pool:
			for l in range(n_lanes):
				pool.submit(lane, l)
			for a in range(0, n_items, group):
				finish(a, [futs[i].result() for i in range(a, min(a + group, n_items))])

	def _find_pipelined(self, queries, in_flight, progress):
		"""find_many, one vk_query per query: the local part on up to `in_flight` handles, the exchange (sharded) in query order"""
		filtered = any(self._filter_masks(q.options) is not None for q in queries)
		handles = self._handles(1 if (filtered or in_flight < 2) else in_flight)
		results = [None] * len(queries)
		done = [0]

		def work(i, l):
			return self._find_local(queries[i], corpus=handles[l] if not filtered else None)

		def finish(first, locals_):
			merged = self._merge_ranks(locals_)
			for j, (loc, top) in enumerate(zip(locals_, merged)):
				results[first + j] = self._finish_find(queries[first + j], loc, top)
			done[0] += len(locals_)
			if progress:
				progress(done[0] / len(queries))
		self._in_order(len(queries), len(handles), work, finish, group=4 if self._shard is not None else 1)
		return results

	def _batch_plan(self, queries, options):
		"""chunks of queries that can go to the backend in one call each, or None"""
		if (self._filter_masks(options) is not None or options.get("debug") is not None
				or not hasattr(self._corpus, "query_batch") or not self._embedding.is_contextual or len(queries) < 2):
			return None
		args, _ = self._backend_args(queries[0].options)
		if "tag_weighted" in args or args["submatch_weight"] != 0.0:
			return None
		alg = args["algorithm"]
		if alg == core.VK_ALG_ALIGN:
			per_call = 16
		elif alg == core.VK_ALG_RWMD and not args.get("wmd_full"):
			per_call = 256
		else:
			return None   # exact transport: per query (bound pass + solver rounds)
		return [range(a, min(a + per_call, len(queries))) for a in range(0, len(queries), per_call)]

	def _find_batches(self, queries, batches, in_flight, progress=None):
		"""find_many through vk_query_batch: the chunks of `batches` on up to two handles of the resident corpus; a sharded index
		exchanges the result sets of a chunk in one all-gather (config 4 sharded: each rank's GEMM, one all-gather of 256 x k records)"""
		emb = self._embedding
		args, gaps = self._backend_args(queries[0].options)
		prepared = [q.prepare(self._nlp) for q in queries]
		handles = self._handles(max(1, min(2, in_flight, len(batches))))
		results = [None] * len(queries)
		k = args["max_matches"]
		done = [0]

		def work(b, l):
			"""local result sets of chunk b: (indices of its non-empty queries, their TopKs, aborted)"""
			idx = [i for i in batches[b] if len(prepared[i]) > 0]
			if not idx and self._shard is None:
				return idx, [], False
			qvs = [emb.encode_tokens(prepared[i].tokens) for i in idx]
			try:
				tops = handles[l].query_batch([np.ascontiguousarray(qv.unmodified, dtype=np.float32) for qv in qvs], q_normalize=True,
					boost=self._dev_boost, want_flow=True, abort_flag=queries[idx[0]]._abort, **args) if idx else []
			except core.VkError as e:
				if e.status != core.VK_ERR_ABORTED:
					raise
				# Query.abort: no matches -- on a sharded index this rank still joins the exchange (with empty result sets and the
				# flag raised), or the ranks that did not see the flag in time would wait for it in the collective forever
				return idx, [self._empty_top(len(prepared[i]), args) for i in idx], True
			return idx, tops, False

		def finish(b, out):
			(idx, tops, aborted), = out
			merged = self._merge_ranks([dict(top=t, aborted=aborted, args=args) for t in tops])
			for i in batches[b]:
				results[i] = []
			for i, top in zip(idx, merged):
				if top is not None:
					results[i] = self._matches_from_topk(prepared[i], top, gaps, args, None, None, None)   # (magnitudes: WRD only, which does not share calls)
			done[0] += len(batches[b])
			if progress:
				progress(done[0] / len(queries))
		self._in_order(len(batches), len(handles), work, finish)
		return results

	def _empty_top(self, len_t, args):
		return core.TopK(max(1, args["max_matches"]), len_t)

	def _merge_ranks(self, locals_):
		"""local results (dicts of _find_local / _find_batches: top, aborted, args) -> the result sets `find` goes on with, None
		for an aborted query.  One GPU: the local ones.  Sharded: ResultSet.extend across the ranks (result_set.h:70-93) -- ONE
		all-gather of the k-record result sets of all the queries handed over (local slice ids -> global), every rank ending
		with the same sets; the similarity rows / plans of transport winners follow in one all-reduce for the merged winners
		only (shards.rows_allreduce), so that flows are stated as on one GPU.  A rank whose query was aborted joins with an
		empty set and a flag: the query then yields no matches on any rank."""
		live = [x for x in locals_ if x is not None]
		if self._shard is None or not live:
			return [None if (x is None or x["aborted"]) else x["top"] for x in locals_]
		from vectorian_amd import shards
		args = live[0]["args"]
		k = args["max_matches"]
		# one exchange per record size (a record holds the query's columns rounded up to 16: queries of 5 and of 20 tokens
		# travel apart), in the order of the sizes -- the same on every rank
		by_size = {}
		for i, x in enumerate(live):
			by_size.setdefault(shards._layout(x["top"].len_t), []).append(i)
		done = [None] * len(live)
		for size in sorted(by_size):
			part = [live[i] for i in by_size[size]]
			tops = [x["top"] for x in part]
			h = shards.allgather_start(tops, self._slice_off, k, group=self._group, device=self._xdev,
				flags=[shards.FLAG_ABORTED if x["aborted"] else 0 for x in part])
			merged = shards.allgather_finish(h)
			# (every rank must take the same decision: by the algorithm, not by what this rank's sets happen to hold -- an aborted
			# or empty local set has no rows)
			transport = args.get("algorithm", core.VK_ALG_ALIGN) != core.VK_ALG_ALIGN
			with_rows = [i for i, x in enumerate(part) if transport or x.get("hook") is not None]   # (a debug hook asks for the rows of alignments too)
			if with_rows:
				lens = [self._slice_end[merged[i].sentence[:merged[i].n]] - self._slice_start[merged[i].sentence[:merged[i].n]] for i in with_rows]
				exact = args.get("algorithm") == core.VK_ALG_WRD or bool(args.get("wmd_full"))
				shards.rows_allreduce([tops[i] for i in with_rows], [merged[i] for i in with_rows], self._slice_off, self._n_local, lens,
					group=self._group, device=self._xdev, with_plan=exact)
			for i, m, f in zip(by_size[size], merged, h["flags_out"]):
				done[i] = None if (f & shards.FLAG_ABORTED) else m
		out = iter(done)
		return [None if x is None else next(out) for x in locals_]

	def _find(self, query, progress=None, corpus=None):
		local = self._find_local(query, corpus=corpus)
		if progress and local is not None:
			progress(self._n_local_tokens / max(1, self._n_tokens))   # this process's tokens are scored (one GPU: all of them)
		top, = self._merge_ranks([local])
		matches = self._finish_find(query, local, top)
		if progress:
			progress(1.0)
		return matches

	def _find_local(self, query, corpus=None):
		"""this process's part of a search: one vk_query against the resident corpus.  Returns None for an empty query, else a
		dict with the local result set (`aborted`: Query.abort was seen -- no matches)"""
		p_query = query.prepare(self._nlp)
		if len(p_query) == 0:
			return None
		args, gaps = self._backend_args(query.options)
		tw = args.pop("tag_weighted", None)
		if tw is not None:
			# parse_tag_weights (vectorian/core/cpp/match/instantiate.cpp:10-38): weight by Penn tag, default 1
			if not self._has_pos:
				raise RuntimeError("tag-weighted similarity needs documents with pos / tags")
			args["tag_weights"] = np.array([float(tw["tag_weights"].get(t, 1.0)) for t in p_query.tags], dtype=np.float32)
			args["q_pos"] = np.array([self.session.pos_code(x) if x is not None else 0 for x in p_query.pos], dtype=np.int8)
			args["pos_mismatch_penalty"] = tw["pos_mismatch_penalty"]
			args["similarity_threshold"] = tw["similarity_threshold"]
		emb = self._embedding
		qv = emb.encode_tokens(p_query.tokens)
		masks = self._filter_masks(query.options)
		if corpus is None:
			corpus = self._filtered_corpus(masks) if masks else self._corpus
		q_tag_codes = None
		if tw is not None and self._has_tags:
			# tag-weighted transport: the vocabulary of the bags of words is keyed by (token id, tag) (TaggedTokenFactory,
			# vectorian/core/cpp/alignment/bow.h:150-176).  The device works on positions, which is the same problem as long as
			# equal (id, tag) pairs have equal similarity rows, i.e. the universal POS is a function of the fine-grained tag (as
			# in spaCy's tag map); the host states the flows of the winners over that vocabulary.
			q_tag_codes = np.array([self.session.tag_code(t) if t is not None else 0 for t in p_query.tags], dtype=np.int8)
			if getattr(corpus, "takes_q_tags", False):
				args["q_tags"] = q_tag_codes
		hook = query.options.get("debug")
		if hook is not None and not callable(hook):
			raise TypeError("debug must be callable: hook(name, data)")   # query.cpp:73-75 casts to a py::object it later calls
		call = dict(args)
		if hook is not None:
			call["want_rows"] = True
		call["abort_flag"] = query._abort
		if emb.is_static:
			call["q_token_ids"] = p_query.token_ids
		aborted = False
		try:
			top = corpus.query(qv.unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, **call)
		except core.VkError as e:
			if e.status != core.VK_ERR_ABORTED:   # Query.abort: the matcher loop ends (match/matcher_impl.h:105), nothing was added
				raise
			top, aborted = self._empty_top(len(p_query), args), True
		call.pop("abort_flag")
		return dict(p_query=p_query, top=top, aborted=aborted, args=args, call=call, gaps=gaps, qv=qv, masks=masks, q_tag_codes=q_tag_codes,
			hook=hook, corpus=corpus)

	def _finish_find(self, query, local, top):
		"""matches of a (merged) result set; the debug hook"""
		if local is None or top is None:
			return []
		args, p_query = local["args"], local["p_query"]
		qmag = np.asarray(local["qv"].magnitudes, dtype=np.float32) if args.get("algorithm") == core.VK_ALG_WRD else None   # masses of the WRD flow
		matches = self._matches_from_topk(p_query, top, local["gaps"], args, qmag, local["masks"], local["q_tag_codes"])
		hook = local["hook"]
		if hook is not None:
			if getattr(hook, "all_slices", False):
				self._call_debug_hook_all_slices(hook, local, matches)
			else:
				self._call_debug_hook(hook, p_query, top, matches, args)
		return matches

	def _call_debug_hook(self, hook, p_query, top, matches, args):
		"""the reference calls hook(name, data) for EVERY slice it scores (call_debug_hook, metric/alignment.h:145-173: slice,
		similarity [len_s x len_t], flow, score = the aligner's score; WMD: 'alignment/word-movers-distance/make' with score and
		worst_score, :600-607).  The scoring kernel keeps no per-slice matrices; by default the hook is called for the k winners, best
		first, with the same keys (`similarity` is None for winners longer than the rows the backend returned).  debug =
		AllSlices(hook) walks every slice instead (_call_debug_hook_all_slices)."""
		alg = args.get("algorithm", core.VK_ALG_ALIGN)
		worst = float(matches[-1].score) if len(matches) >= args["max_matches"] else float(args["min_score"])
		for i, m in enumerate(matches):
			if alg == core.VK_ALG_WRD or (alg == core.VK_ALG_RWMD and args.get("wmd_full")):
				data = self._solver_debug_data(p_query, top, i, m, args)
				if data is not None:
					hook("alignment/word-rotators-distance/solver" if alg == core.VK_ALG_WRD else "alignment/word-movers-distance/solver", data)
			if alg == core.VK_ALG_RWMD:
				hook("alignment/word-movers-distance/make", {"score": m.score, "worst_score": worst, "slice": m.slice_id, "flow": m.flow})
				continue
			if alg == core.VK_ALG_WRD:
				continue
			sim = None
			if getattr(top, "sim_rows", None) is not None and m._len_s <= _rows_room(top, i):
				sim = top.sim_rows[i][:m._len_s if m._index_map is None else len(m._index_map), :len(p_query)].copy()
			hook("alignment", {"slice": m.slice_id, "similarity": sim, "flow": m.flow, "score": m.raw_score})

	def _token_magnitudes(self, g):
		"""|x| of the tokens of slice g, as Slice::magnitude_s hands them to WRD (metric/contextual.cpp:49-54, metric/static.cpp:69-73)"""
		a, b = int(self._slice_start[g]), int(self._slice_end[g])
		emb = self._embedding
		if emb.is_static:
			if getattr(self, "_vocab_mag", None) is None:
				self._vocab_mag = np.asarray(emb.encode_tokens(self.session.vocab.tokens).magnitudes, dtype=np.float32)
			return self._vocab_mag[self._token_ids[a:b]]
		from vectorian_amd.embedding import Vectors
		doc_base = 0
		for doc in self.session.documents:
			if a < doc_base + doc.n_tokens:
				return np.asarray(Vectors(doc.contextual_vectors(emb.name)[a - doc_base:b - doc_base]).magnitudes, dtype=np.float32)
			doc_base += doc.n_tokens
		return None

	def _solver_debug_data(self, p_query, top, i, m, args):
		"""what the exact solvers hand the debug hook for a slice (WRD::call_debug_hook, vectorian/core/cpp/alignment/wrd.h:31-59;
		FullSolver::call_debug_hook, alignment/wmd.h:147-181), stated for a WINNER from the similarity rows and the optimal plan the
		backend returned: tokens of both sides, masses, the distance matrix over the joint problem, the plan G and its cost"""
		if getattr(top, "sim_rows", None) is None or getattr(top, "plan", None) is None or m._index_map is not None:
			return None
		len_s, len_t = m._len_s, len(p_query)
		if len_s > _rows_room(top, i):
			return None
		g = int(m._w.sent[m._i])
		a = int(self._slice_start[g])
		doc_tokens = m.prepared_doc.tokens
		S = top.sim_rows[i][:len_s, :len_t]
		G_ts = top.plan[i][:len_t, :len_s]
		ids_s = self._token_ids[a:a + len_s].tolist() if self._token_ids is not None else list(range(len_s))
		data = {
			"s": {"id": ids_s, "text": list(doc_tokens[m._token_at:m._token_at + len_s])},
			"t": {"id": [int(x) for x in p_query.token_ids], "text": list(p_query.tokens)}}
		n = len_s + len_t
		D = np.ones((n, n), dtype=np.float32)
		D[:len_t, len_t:] = np.maximum(0.0, 1.0 - S.T)
		G = np.zeros((n, n), dtype=np.float32)
		G[:len_t, len_t:] = G_ts
		if args.get("algorithm") == core.VK_ALG_WRD:
			mag_s, mag_t = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
			ms = self._token_magnitudes(g)
			if ms is None:
				return None
			mag_t[:len_t] = np.asarray(self._embedding.encode_tokens(p_query.tokens).magnitudes, dtype=np.float32)
			mag_s[len_t:] = ms
			if args.get("wrd_normalize", True):
				mag_t /= mag_t.sum()
				mag_s /= mag_s.sum()
			data.update(mag_s=mag_s, mag_t=mag_t, D=D, elapsed_microseconds=0,
				solution={"G": G, "cost": float((D * G).sum()), "type": "optimal"})
			return data
		# full WMD over positions (every position its own vocabulary entry on the device; the host states flows over the joint vocabulary)
		nbow = args["rwmd"][2]
		data.update(bow_s=np.full(len_s, 1.0 / len_s if nbow else 1.0, dtype=np.float32), bow_t=np.full(len_t, 1.0 / len_t if nbow else 1.0, dtype=np.float32),
			D=D, G=G, flow_by_pos=m.flow["flow"], dist_by_pos=m.flow["dist"], score=m.raw_score)
		return data

	def _call_debug_hook_all_slices(self, hook, local, matches):
		"""debug = AllSlices(hook): the hook contract of the reference in full -- one call per slice this process scores, in slice
		order, with the reference's keys.  Alignments ('alignment': slice, similarity, flow, score; metric/alignment.h:145-173): the
		slices are stated `hook.chunk` at a time by the traceback kernel (vk_query_desc.only_slices: aligner score, mapping, edge
		similarities and the similarity rows, canonical arithmetic), whatever their score.  Relaxed WMD
		('alignment/word-movers-distance/make': score, worst_score; :600-607): the score of every slice restated from its canonical
		similarity rows (only_slices again: the reference's floats), with the worst score of a result set filled in slice order as
		upstream fills it.  Exact transports: every slice solved, the solver's hook per slice (tokens, masses, distance matrix, plan,
		cost).  Opt-in and slow (a Python call per slice); a sharded index walks its own slices on every rank.  A submatch weight does
		not change what the hook of an alignment is handed (the aligner's score, not Score::value)."""
		args, p_query, corpus = local["args"], local["p_query"], local["corpus"]
		alg = args.get("algorithm", core.VK_ALG_ALIGN)
		n_loc, off, len_t = self._n_local, self._slice_off, len(p_query)
		if alg == core.VK_ALG_RWMD and not args.get("wmd_full"):
			import heapq
			# the scores of all slices as the backend states winners: restated from canonical similarity rows in the reference's order
			# of operations (vk_query_desc.only_slices, `hook.chunk` slices per call) -- the floats upstream hands its hook; slices
			# longer than the rows a call returns keep the scoring pass's value
			scores = np.array(corpus.last_scores(), dtype=np.float32)
			call, masks, lens_all = dict(local["call"]), local["masks"], self._slice_end[off:off + n_loc] - self._slice_start[off:off + n_loc]
			for a in range(0, n_loc, max(1, int(getattr(hook, "chunk", 512)))):
				ids = np.arange(a, min(n_loc, a + max(1, int(getattr(hook, "chunk", 512)))), dtype=np.int64)
				ids = ids[(lens_all[ids] > 0) & np.isfinite(scores[ids])]
				if len(ids) == 0:
					continue
				top = corpus.query(local["qv"].unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, only_slices=ids, **call)
				stated = np.isfinite(top.score[:top.n])
				scores[ids[:top.n][stated]] = top.score[:top.n][stated]
			heap, k, floor = [], args["max_matches"], float(args["min_score"])
			for g in range(n_loc):
				sc = float(scores[g])
				if not np.isfinite(sc):
					continue   # empty slice: Spans::iterate skips it (document.h:160-162)
				worst = heap[0] if len(heap) >= k else floor
				hook("alignment/word-movers-distance/make", {"score": sc, "worst_score": worst, "slice": self._slice_id[off + g]})
				if sc > worst:
					heapq.heappush(heap, sc) if len(heap) < k else heapq.heapreplace(heap, sc)
			return
		if alg == core.VK_ALG_WRD or (alg == core.VK_ALG_RWMD and args.get("wmd_full")):
			# exact transports: every slice solved (only_slices: no bound pass, nothing pruned), the solver's hook per slice
			# (WRD::call_debug_hook, wrd.h:31-59; FullSolver::call_debug_hook, wmd.h:147-181) and, full WMD, 'make' with the worst
			# score of a result set filled in slice order
			import heapq
			call, masks = dict(local["call"]), local["masks"]
			qmag = np.asarray(local["qv"].magnitudes, dtype=np.float32) if alg == core.VK_ALG_WRD else None
			lens_all = self._slice_end[off:off + n_loc] - self._slice_start[off:off + n_loc]
			heap, k, floor = [], args["max_matches"], float(args["min_score"])
			name = "alignment/word-rotators-distance/solver" if alg == core.VK_ALG_WRD else "alignment/word-movers-distance/solver"
			for a in range(0, n_loc, max(1, int(getattr(hook, "chunk", 512)))):
				ids = np.arange(a, min(n_loc, a + max(1, int(getattr(hook, "chunk", 512)))), dtype=np.int64)
				ids = ids[lens_all[ids] > 0]
				if len(ids) == 0:
					continue
				top = corpus.query(local["qv"].unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, only_slices=ids, **call)
				chunk = self._matches_from_topk(p_query, top, local["gaps"], args, qmag, masks, local["q_tag_codes"])
				for i, m in enumerate(chunk):
					if not np.isfinite(m.score):
						continue   # every token filtered out: the slice is not scored
					data = self._solver_debug_data(p_query, top, i, m, args)
					if data is not None:
						hook(name, data)
					if alg == core.VK_ALG_RWMD:
						worst = heap[0] if len(heap) >= k else floor
						hook("alignment/word-movers-distance/make", {"score": m.score, "worst_score": worst, "slice": m.slice_id, "flow": m.flow})
						if m.score > worst:
							heapq.heappush(heap, m.score) if len(heap) < k else heapq.heapreplace(heap, m.score)
			return
		if alg != core.VK_ALG_ALIGN:
			return self._call_debug_hook(hook, p_query, local["top"], matches, args)
		call = dict(local["call"], want_rows=True)
		masks = local["masks"]
		for a in range(0, n_loc, max(1, int(getattr(hook, "chunk", 512)))):
			ids = np.arange(a, min(n_loc, a + max(1, int(getattr(hook, "chunk", 512)))), dtype=np.int64)
			lens = self._slice_end[off + ids] - self._slice_start[off + ids]
			ids = ids[lens > 0]   # Spans::iterate skips empty slices
			if len(ids) == 0:
				continue
			top = corpus.query(local["qv"].unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, only_slices=ids, **call)
			for i in range(top.n):
				g = off + int(ids[i])
				len_s = int(self._slice_end[g] - self._slice_start[g]) if not masks else len(self._index_map(g, masks))
				if len_s < 1:
					continue   # every token filtered out: the slice is not scored (FilteredSliceFactory, slice/static.h:366-416)
				target = top.mapping[i].astype(np.int16)
				matched = target >= 0
				flow = {"type": "injective", "target": target, "flow": matched.astype(np.float32),
					"dist": np.where(matched, 1.0 - top.edge_sim[i], 1.0).astype(np.float32)}
				sim = top.sim_rows[i][:len_s, :len_t].copy() if len_s <= top.sim_rows.shape[1] else None
				hook("alignment", {"slice": self._slice_id[g], "similarity": sim, "flow": flow, "score": float(top.raw_score[i])})

	def _filter_masks(self, options):
		"""pos_filter / tag_filter: lists of POS / tag names whose tokens are dropped from every slice for this query
		(Query::make_token_filter, vectorian/core/cpp/query.cpp:220-228; parse_filter_mask, query.h:33-55).
		Returns (pos_mask, tag_mask) or None."""
		masks = []
		for name, lookup, have in (("pos_filter", self.session.pos_id, self._has_pos), ("tag_filter", self.session.tag_id, self._has_tags)):
			m = 0
			for x in options.get(name) or ():
				i = lookup(str(x)) if have else -1
				if i < 0 or i > 63:
					raise RuntimeError(f"illegal value {x} for {name}")   # query.h:45-49
				m |= 1 << i
			masks.append(m)
		return tuple(masks) if any(masks) else None

	def _filtered_corpus(self, masks):
		c = self._filtered.pop(masks, None)
		if c is None:
			c = self._corpus.filtered(*masks)
			while len(self._filtered) >= 2:
				self._filtered.popitem(last=False)[1].close()
		self._filtered[masks] = c
		return c

	def _index_map(self, g, masks):
		"""positions of the tokens of slice g that pass the filter (the index_map of Flow::py_regions, flow.cpp:49-60)"""
		a, b = int(self._slice_start[g]), int(self._slice_end[g])
		drop = np.zeros(b - a, dtype=bool)
		for mask, codes in zip(masks, (self._pos_codes, self._tag_codes)):
			if mask and codes is not None:
				cd = codes[a:b].astype(np.int64)
				bits = np.array([(mask >> b) & 1 for b in range(64)], dtype=bool)   # codes 0..63 have a bit (query.h:13-14)
				drop |= (cd >= 0) & (cd < 64) & bits[np.clip(cd, 0, 63)]
		return np.nonzero(~drop)[0]

	def _transport_flow(self, p_query, top, i, g, args, qmag, index_map=None, q_tag_codes=None, span=None):
		"""flow of winner i of a transport query, stated from the similarity rows / plan the backend returned
		(a callable: HipMatch.flow evaluates it when asked)"""
		if getattr(top, "sim_rows", None) is None:
			return None
		a, b = span if span is not None else (int(self._slice_start[g]), int(self._slice_end[g]))
		len_s, len_t = (b - a if index_map is None else len(index_map)), len(p_query)
		if len_s > _rows_room(top, i):   # rows per winner the backend was given room for (the corpus's longest slice on the HIP backend)
			return None
		alg = args["algorithm"]
		token_ids, tag_codes = self._token_ids, self._tag_codes

		def state():
			S = top.sim_rows[i][:len_s, :len_t].copy()
			ids_s = token_ids[a:b] if token_ids is not None else None
			if ids_s is not None and index_map is not None:
				ids_s = ids_s[index_map]
			ids_t = p_query.token_ids if token_ids is not None else None
			if ids_s is not None and q_tag_codes is not None and tag_codes is not None:
				# tag-weighted: vocabulary entries are (token id, tag) pairs
				tags_s = tag_codes[a:b] if index_map is None else tag_codes[a:b][index_map]
				ids_s = np.asarray(ids_s, dtype=np.int64) * 256 + (np.asarray(tags_s, dtype=np.int64) & 255)
				ids_t = np.asarray(ids_t, dtype=np.int64) * 256 + (np.asarray(q_tag_codes, dtype=np.int64) & 255)
			if alg == core.VK_ALG_WRD:
				mass = qmag / qmag.sum() if args.get("wrd_normalize", True) else qmag
				return dense_flow(S, top.plan[i][:len_t, :len_s].copy(), None, None, mass)   # WRD works on positions (wrd.h:91-109)
			injective, symmetric, nbow = args["rwmd"]
			if args.get("wmd_full"):
				unit = 1.0 / len_t if nbow else 1.0
				return dense_flow(S, top.plan[i][:len_t, :len_s].copy(), ids_s, ids_t, np.full(len_t, unit, dtype=np.float32))
			return rwmd_sparse_flow(S, ids_s, ids_t, injective, symmetric, nbow)
		return state

	def _matches_from_topk(self, p_query, top, gaps, args=None, qmag=None, masks=None, q_tag_codes=None):
		"""the matches of a result set: one shared `_Winners` and a two-word object per winner; everything else on access"""
		if top.n == 0:
			return []
		w = _Winners(self, p_query, top, gaps, args, qmag, masks, q_tag_codes)
		return [HipMatch(w, i) for i in range(w.n)]

	def close(self):
		for c in list(self._filtered.values()) + self._views:
			c.close()
		self._filtered.clear()
		self._views = []
		self._corpus.close()


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


class HipSpanEncoderIndex(Index):
	"""Brute-force cosine search over ONE embedding per span: what SpanEncoderIndex / FaissCosineIndex('Flat')
	do (vectorian/index.py:679-810), as a matrix-vector product + bounded result set on the GPU.  Realised on
	the same kernels: a corpus of one-token slices, a one-token query, score = clip(cosine)."""

	def __init__(self, partition, embedding, span_sim, nlp=None, vectors=None, device=0, corpus_factory=None):
		super().__init__(partition, span_sim)
		self._embedding = embedding
		self._nlp = nlp
		session = self.session
		step, level = partition.window_step, partition.level
		self._slice_doc, self._slice_id = [], []
		for di, doc in enumerate(session.documents):
			for sid in range(0, doc.n_spans(level), step):
				self._slice_doc.append(di)
				self._slice_id.append(sid)
		n = len(self._slice_doc)
		if vectors is None:
			size = partition.window_size
			texts = [" ".join(session.documents[di].span_tokens(level, sid, size)) for di, sid in zip(self._slice_doc, self._slice_id)]
			vectors = embedding.encode(texts)
		vectors = np.ascontiguousarray(vectors, dtype=np.float32)
		if vectors.shape != (n, embedding.dimension):
			raise ValueError(f"expected {(n, embedding.dimension)} span vectors, got {vectors.shape}")
		make = corpus_factory or core.Corpus
		self._corpus = make(layout=core.VK_LAYOUT_CONTEXTUAL, d=embedding.dimension, n_tokens=n, n_sentences=n, device=device)
		self._corpus.append_vectors(vectors, normalize=True)
		self._corpus.set_sentences(np.arange(n + 1, dtype=np.int64))
		self._corpus.finalize()

	def _find(self, query, progress=None):
		qv = self._embedding.encode([query.text])
		if qv.shape[0] != 1:
			raise RuntimeError("query produced more than one embedding")
		top = self._corpus.query(qv, q_normalize=True, locality=core.Locality.LOCAL, gap_s=0.0, gap_t=0.0,
			max_matches=int(query.options.get("max_matches", 100)), min_score=float(query.options.get("min_score", 0.0)),
			want_flow=False)
		if progress:
			progress(1.0)
		matches = []
		for i in range(top.n):
			g = int(top.sentence[i])
			doc = self.session.documents[self._slice_doc[g]]
			text = " ".join(doc.span_tokens(self.partition.level, self._slice_id[g], self.partition.window_size))
			matches.append(PyMatch(self, query, doc, self._slice_id[g], float(top.score[i]), self._sim.name,
				regions=[Region(s=text, match=None, gap_penalty=0)], level="span"))
		return matches

	def close(self):
		self._corpus.close()
