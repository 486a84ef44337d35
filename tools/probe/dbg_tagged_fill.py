"""probe: tag-weighted 1:n RWMD over the static layout, every slice's score against the oracle's"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from vectorian_amd import core as hip, synth
from oracle import vk_oracle as oracle
hip.init(0)
len_t = int(sys.argv[1]) if len(sys.argv) > 1 else 40
V, d = 300, 64
corpus = synth.make_static_corpus(400, 1, 40, V, d, seed=41)
rng = np.random.default_rng(42)
E = (corpus["E"] * rng.lognormal(0, 0.3, size=(V, 1))).astype(np.float32)
Eb, emag = oracle.normalize_rows_bf16(E)
off, ids = corpus["sent_off"], corpus["tok_id"]
tag_s = rng.integers(1, 9, size=len(ids)).astype(np.int8)
pos_s = (tag_s % 3 + 1).astype(np.int8)
c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=len(ids), n_sentences=len(off) - 1, vocab_size=V, keep_magnitudes=True)
c.append_vectors(E, normalize=True); c.set_token_ids(ids); c.set_sentences(off); c.set_token_pos(pos_s); c.set_token_tags(tag_s); c.finalize()
for rep in range(2):
	q_ids = rng.integers(0, 40, size=len_t).astype(np.int32)
	q_tag = rng.integers(1, 9, size=len_t).astype(np.int8)
	q_pos = (q_tag % 3 + 1).astype(np.int8)
	tw = np.array([0.5, 1.0, 3.0, 1.5, 0.75, 2.0, 1.0, 0.25, 1.25], dtype=np.float32)[q_tag]
	for flags in ((True, True, True), (True, False, True), (True, False, False), (False, True, True)):
		kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.25, similarity_threshold=0.05, max_matches=10, min_score=-1.0, rwmd=flags)
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids, Q_mag=emag[q_ids],
			pos_s=pos_s, tag_s=tag_s, q_tag=q_tag, algorithm=oracle.ALG_RWMD, want_all_scores=True, **kw)
		got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, algorithm=hip.VK_ALG_RWMD, q_tags=q_tag, **kw)
		sc = c.last_scores()
		diff = np.abs(sc - ref["all_scores"])
		bad = np.nonzero(diff > 2e-5)[0]
		print("rep", rep, flags, "bad slices", len(bad), "max diff", float(diff.max()))
		for s in bad[:3]:
			a, b = int(off[s]), int(off[s + 1])
			dup_q = [(int(i), int(t)) for i, t in zip(q_ids, q_tag)]
			print("  slice", s, "len", b - a, "gpu", sc[s], "oracle", ref["all_scores"][s], "slice keys shared with query:",
				sorted(set(zip(ids[a:b].tolist(), tag_s[a:b].tolist())) & set(dup_q)), "ids shared:", sorted(set(ids[a:b].tolist()) & set(q_ids.tolist())))
