"""Flows of transport winners, stated on the host from what the backend returned for them (similarity rows, optimal plans):
SparseFlow of the relaxed WMD (vectorian/core/cpp/alignment/wmd.h:392-408), DenseFlow of the exact transports (wmd.h:228-248,
wrd.h:120-135), over the joint vocabularies of alignment/bow.h:204-333."""

import numpy as np

from vectorian_amd import core


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



def transport_flow(index, p_query, top, i, g, args, qmag, index_map=None, q_tag_codes=None, span=None):
	"""flow of winner i of a transport query, stated from the similarity rows / plan the backend returned
	(a callable: HipMatch.flow evaluates it when asked)"""
	if getattr(top, "sim_rows", None) is None:
		return None
	a, b = span if span is not None else (int(index._slice_start[g]), int(index._slice_end[g]))
	len_s, len_t = (b - a if index_map is None else len(index_map)), len(p_query)
	if len_s > _rows_room(top, i):   # rows per winner the backend was given room for (the corpus's longest slice on the HIP backend)
		return None
	alg = args["algorithm"]
	token_ids, tag_codes = index._token_ids, index._tag_codes

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
