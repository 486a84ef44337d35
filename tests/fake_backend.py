"""Test double for `vectorian_amd.core.Corpus`: same interface, scores computed by the CPU
oracle.  Lives in tests/ only -- it lets the host-side plumbing (Session / Index / shards)
run in the no-GPU tier; the product never imports it."""

import threading

import numpy as np

from oracle import vk_oracle as vo
from vectorian_amd import core, synth


def _gap(g, n=core.VK_MAX_SENT_LEN + 1):   # (65 entries until round 3: a gap over more than 64 tokens of a long slice cost +inf in the double)
	if hasattr(g, "to_special_case"):
		sp = g.to_special_case()
		if "linear" in sp:
			return ("linear", sp["linear"])
		if "affine" in sp:
			return ("affine",) + tuple(sp["affine"])
		return ("table", g.costs(n))
	return g


class OracleCorpus:
	takes_q_tags = True   # tag-weighted transport: the oracle keys the bags of words by (token id, tag), as the reference does

	def __init__(self, *, layout, d, n_tokens, n_sentences, vocab_size=0, keep_magnitudes=False, device=None, precision="bf16"):
		self.layout, self.d = layout, d
		self.precision = precision
		self.n_tokens, self.n_sentences, self.vocab_size = n_tokens, n_sentences, vocab_size
		self._rows, self._mags = [], []
		self._ids = None
		self._pos = None
		self._tags = None
		self._off = None
		self._all = None
		self.lock = threading.RLock()   # as core.Corpus: one call at a time per handle (view() returns this same object)

	def append_vectors(self, rows, normalize=True):
		rows = np.ascontiguousarray(rows)
		if rows.dtype == np.uint16:
			rows = synth.bf16_bits_to_f32(rows)
		if self.precision == "f32":
			b, m = (vo.normalize_rows(rows), vo.magnitudes(rows)) if normalize else (rows.astype(np.float32), np.ones(len(rows), np.float32))
		elif normalize:
			b, m = vo.normalize_rows_bf16(rows)
		else:
			b, m = synth.to_bf16_bits(rows), np.ones(len(rows), np.float32)
		self._rows.append(b)
		self._mags.append(m)

	def _longest(self):
		"""tokens of the longest slice (a gap table covers it; whole documents as slices: more than VK_MAX_SENT_LEN)"""
		ends = self._end if self._end is not None else self._off[1:]
		return int((ends - self._off[:len(ends)]).max(initial=0))

	def set_token_ids(self, ids):
		self._ids = np.ascontiguousarray(ids, dtype=np.int32)

	def set_token_pos(self, pos):
		self._pos = np.ascontiguousarray(pos, dtype=np.int8)

	def set_token_tags(self, tags):
		self._tags = np.ascontiguousarray(tags, dtype=np.int8)

	def filtered(self, pos_mask=0, tag_mask=0):
		"""the corpus without the tokens the filter drops, slices re-indexed: restated on the host, array by array"""
		n = self.n_tokens
		drop = np.zeros(n, dtype=bool)
		for mask, codes in ((pos_mask, self._pos), (tag_mask, self._tags)):
			if mask:
				if codes is None:
					raise core.VkError(4, "filter needs token codes")
				cd = codes.astype(np.int64)
				bits = np.array([(int(mask) >> b) & 1 for b in range(64)], dtype=bool)
				drop |= (cd >= 0) & (cd < 64) & bits[np.clip(cd, 0, 63)]
		keep = ~drop
		new_index = np.concatenate(([0], np.cumsum(keep))).astype(np.int64)
		f = OracleCorpus(layout=self.layout, d=self.d, n_tokens=int(keep.sum()), n_sentences=self.n_sentences,
			vocab_size=self.vocab_size, precision=self.precision)
		f._X, f._mag = self._X, self._mag
		if self.layout == core.VK_LAYOUT_STATIC:
			f._ids = self._ids[keep]
		else:
			f._X, f._mag = self._X[keep], self._mag[keep]
		f._pos = None if self._pos is None else self._pos[keep]
		f._tags = None if self._tags is None else self._tags[keep]
		f._off = new_index[self._off]
		f._end = None if self._end is None else new_index[self._end]
		return f

	def set_sentences(self, off):
		self._off = np.ascontiguousarray(off, dtype=np.int64)
		self._end = None

	def set_slices(self, start, end):
		self._off = np.ascontiguousarray(start, dtype=np.int64)
		self._end = np.ascontiguousarray(end, dtype=np.int64)

	def finalize(self):
		self._X = np.concatenate(self._rows) if self._rows else np.zeros((0, self.d), np.float32 if self.precision == "f32" else np.uint16)
		self._mag = np.concatenate(self._mags) if self._mags else np.zeros(0, np.float32)

	def query(self, q_vectors, *, locality=0, gap_s=0.0, gap_t=0.0, algorithm=0, q_token_ids=None, q_normalize=True,
			max_matches=10, min_score=0.0, boost=None, want_flow=True, submatch_weight=0.0, bidirectional=False,
			rwmd=(True, True, True), wrd_normalize=True, tag_weights=None, q_pos=None, pos_mismatch_penalty=0.0,
			similarity_threshold=0.0, wmd_full=False, q_tags=None, abort_flag=None, want_rows=False, only_slices=None):
		if abort_flag is not None and abort_flag[0]:
			raise core.VkError(core.VK_ERR_ABORTED, "query aborted by the caller")
		if only_slices is not None:
			# vk_query_desc.only_slices: exactly these slices, in this order, whatever their score (the debug hook's walk)
			ids = np.asarray(only_slices, dtype=np.int64)
			sub = OracleCorpus(layout=self.layout, d=self.d, n_tokens=self.n_tokens, n_sentences=len(ids), vocab_size=self.vocab_size, precision=self.precision)
			sub._X, sub._mag, sub._ids, sub._pos, sub._tags = self._X, self._mag, self._ids, self._pos, self._tags
			sub._off = self._off[ids]
			sub._end = (self._end if self._end is not None else self._off[1:])[ids]
			b = None if boost is None else np.ascontiguousarray(np.asarray(boost, dtype=np.float32)[ids])
			empty = (sub._end - sub._off) < 1   # (a filter may have emptied a slice: stated with score -inf, as the HIP backend does)
			if empty.any():
				full = self.query(q_vectors, locality=locality, gap_s=gap_s, gap_t=gap_t, algorithm=algorithm, q_token_ids=q_token_ids, q_normalize=q_normalize,
					max_matches=max_matches, min_score=min_score, boost=boost, want_flow=want_flow, submatch_weight=submatch_weight, rwmd=rwmd,
					wrd_normalize=wrd_normalize, tag_weights=tag_weights, q_pos=q_pos, pos_mismatch_penalty=pos_mismatch_penalty,
					similarity_threshold=similarity_threshold, wmd_full=wmd_full, q_tags=q_tags, want_rows=want_rows, only_slices=ids[~empty]) if (~empty).any() else None
				transport = (algorithm != core.VK_ALG_ALIGN or want_rows) and want_flow
				longest = int((sub._end - sub._off).max())
				t = core.TopK(len(ids), len(np.atleast_2d(q_vectors)), transport=transport, rows=core.winner_rows(longest))
				t.n = len(ids)
				t.score[:], t.raw_score[:] = -np.inf, -np.inf
				t.sentence[:t.n] = ids
				if full is not None:
					at = np.nonzero(~empty)[0]
					for name in ("score", "raw_score", "mapping", "edge_sim") + (("sim_rows", "plan") if getattr(full, "sim_rows", None) is not None and t.sim_rows is not None else ()):
						dst, src = getattr(t, name), getattr(full, name)
						if dst.ndim == 1:
							dst[at] = src[:full.n]
						else:
							sl = tuple(slice(0, min(a_, b_)) for a_, b_ in zip(dst.shape[1:], src.shape[1:]))
							dst[(at,) + sl] = src[(slice(0, full.n),) + sl]
				return t
			t = sub.query(q_vectors, locality=locality, gap_s=gap_s, gap_t=gap_t, algorithm=algorithm, q_token_ids=q_token_ids, q_normalize=q_normalize,
				max_matches=len(ids), min_score=-3.0e38, boost=b, want_flow=want_flow, submatch_weight=submatch_weight, rwmd=rwmd,
				wrd_normalize=wrd_normalize, tag_weights=tag_weights, q_pos=q_pos, pos_mismatch_penalty=pos_mismatch_penalty,
				similarity_threshold=similarity_threshold, wmd_full=wmd_full, q_tags=q_tags, want_rows=want_rows)
			assert t.n == len(ids), (t.n, len(ids))
			back = np.argsort(t.sentence[:t.n], kind="stable")   # position i of the output = slice ids[i]
			for name in ("score", "raw_score", "mapping", "edge_sim") + (("sim_rows", "plan") if getattr(t, "sim_rows", None) is not None else ()):
				arr = getattr(t, name)
				arr[:t.n] = arr[:t.n][back]
			t.sentence[:t.n] = ids
			return t
		q = np.ascontiguousarray(q_vectors)
		if q.dtype == np.uint16:
			q = synth.bf16_bits_to_f32(q)
		q = q.astype(np.float32)
		if self.precision == "f32":
			Qb, qmag = (vo.normalize_rows(q), vo.magnitudes(q)) if q_normalize else (q, np.ones(len(q), np.float32))
		elif q_normalize:
			Qb, qmag = vo.normalize_rows_bf16(q)
		else:
			Qb, qmag = synth.to_bf16_bits(q), np.ones(len(q), np.float32)
		kw = dict(layout=self.layout, d=self.d, sent_off=self._off, sent_end=self._end, Q=Qb, algorithm=algorithm, locality=int(locality),
			gap_s=_gap(gap_s, max(core.VK_MAX_SENT_LEN, self._longest()) + 1), gap_t=_gap(gap_t), max_matches=max_matches, min_score=min_score, boost=boost,
			submatch_weight=submatch_weight, rwmd=rwmd, wrd_normalize=wrd_normalize, want_all_scores=True,
			pos_s=self._pos, tag_weights=tag_weights, q_pos=q_pos, pos_mismatch_penalty=pos_mismatch_penalty,
			similarity_threshold=similarity_threshold, wmd_full=wmd_full)
		if tag_weights is not None and q_tags is not None and self._tags is not None:
			kw.update(tag_s=self._tags, q_tag=q_tags)
		if self.layout == core.VK_LAYOUT_STATIC:
			kw.update(tok_id=self._ids, E=self._X, q_ids=q_token_ids, X_mag=self._mag[self._ids], Q_mag=qmag)
		else:
			kw.update(X=self._X, X_mag=self._mag, Q_mag=qmag)
		r = vo.find(**kw)
		self._all = np.array(r["all_scores"], dtype=np.float32)
		ends = self._end if self._end is not None else self._off[1:]
		self._all[(ends - self._off[:len(ends)]) < 1] = -np.inf   # empty slices carry no score (vk_last_scores: -inf; Spans::iterate skips them)
		transport = (algorithm != core.VK_ALG_ALIGN or want_rows) and want_flow
		# room for the rows / plans of the corpus's longest slice, as the HIP backend's shim makes (core.Corpus._winner_rows)
		all_ends = self._end if self._end is not None else self._off[1:]
		longest = int((all_ends - self._off[:len(all_ends)]).max()) if len(all_ends) else 0
		rows_room = core.winner_rows(longest)
		top = core.TopK(max_matches, len(q), transport=transport, rows=rows_room)
		n = len(r["score"])
		top.n = n
		top.score[:n], top.raw_score[:n], top.sentence[:n] = r["score"], r["raw"], r["sentence"]
		top.mapping[:n] = r["mapping"]
		# edge similarities as the flow kernel reports them
		for i in range(n):
			s = int(r["sentence"][i])
			a, b = int(self._off[s]), int(self._end[s] if self._end is not None else self._off[s + 1])
			if self.precision == "f32":
				rows = self._X[self._ids[a:b]] if self.layout == core.VK_LAYOUT_STATIC else self._X[a:b]
				S = vo.sim_f32(rows, Qb)
				if self.layout == core.VK_LAYOUT_STATIC and q_token_ids is not None:
					for j, qid in enumerate(q_token_ids):
						S[self._ids[a:b] == qid, j] = 1.0
			elif self.layout == core.VK_LAYOUT_STATIC:
				table = vo.sim_table_static_bf16(self._X, Qb, q_token_ids)
				S = table[self._ids[a:b]]
			else:
				S = vo.sim_bf16(self._X[a:b], Qb)
			for j in range(len(q)):
				if r["mapping"][i][j] >= 0:
					top.edge_sim[i, j] = S[r["mapping"][i][j], j]
			if transport and tag_weights is not None:
				# the solvers (and the flows stated from these rows) see the modified similarity (TagWeightedSlice::similarity, slice/static.h:237-264)
				wgt = np.asarray(tag_weights, dtype=np.float32)[None, :] * np.where(
					self._pos[a:b, None] != np.asarray(q_pos, dtype=np.int8)[None, :], (np.float32(1.0) - np.float32(pos_mismatch_penalty)), np.float32(1.0))
				raw = S
				S = (raw * wgt).astype(np.float32)
				S[S <= similarity_threshold] = 0.0
				if self.layout == core.VK_LAYOUT_STATIC and algorithm == core.VK_ALG_RWMD and q_tags is not None and self._tags is not None and q_token_ids is not None:
					# upstream's vocabulary distance matrix is written twice for entries that occur in both documents, the later write
					# wins (wmd.h:121-133): for keys a (slice) < b (query), both in both, the value of (slice's b, query's a) -- this
					# cell's cosine and POS penalty under the tag weight of a (the device: static_vocab_fixup)
					ks = self._ids[a:b].astype(np.int64) * 256 + (self._tags[a:b].astype(np.int64) & 255)
					kt = np.asarray(q_token_ids, dtype=np.int64) * 256 + (np.asarray(q_tags, dtype=np.int64) & 255)
					in_t, in_s = np.isin(ks, kt), np.isin(kt, ks)
					for si in np.nonzero(in_t)[0]:
						ft = int(np.nonzero(kt == ks[si])[0][0])
						for tj in np.nonzero(in_s & (kt > ks[si]))[0]:
							w = np.float32(tag_weights[ft]) * ((np.float32(1.0) - np.float32(pos_mismatch_penalty)) if self._pos[a + si] != q_pos[tj] else np.float32(1.0))
							v = np.float32(raw[si, tj] * w)
							S[si, tj] = 0.0 if v <= similarity_threshold else v
			if transport and b - a <= rows_room:
				# what the HIP backend returns for the host to state transport flows: rows, and the plan of exact transports
				top.sim_rows[i, :b - a, :len(q)] = S
				exact = algorithm == core.VK_ALG_WRD or (algorithm == core.VK_ALG_RWMD and wmd_full)
				if exact:
					if algorithm == core.VK_ALG_WRD:
						ms = self._mag[self._ids[a:b]] if self.layout == core.VK_LAYOUT_STATIC else self._mag[a:b]
						mt = qmag
						if wrd_normalize:
							ms, mt = ms / ms.sum(dtype=np.float32), mt / mt.sum(dtype=np.float32)
					else:
						nb = rwmd[2]
						ms = np.full(b - a, 1.0 / (b - a) if nb else 1.0, np.float32)
						mt = np.full(len(q), 1.0 / len(q) if nb else 1.0, np.float32)
					_, G = vo.emd(mt, ms, np.maximum(1.0 - S.T, 0.0))
					top.plan[i, :len(q), :b - a] = G
		return top

	def query_batch(self, queries, token_ids=None, **options):
		"""vk_query_batch's contract: n_queries calls of vk_query with common options"""
		self.batch_calls = getattr(self, "batch_calls", 0) + 1
		if token_ids is not None:
			return [self.query(q, **dict(options, q_token_ids=t)) for q, t in zip(queries, token_ids)]
		return [self.query(q, **options) for q in queries]

	def view(self):
		return self          # the double keeps no per-query state a second handle would need

	def last_scores(self):
		return self._all

	def close(self):
		pass
