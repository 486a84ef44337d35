"""Probe kept from round 2: the 1:n RWMD (rwmd_fill32 / rwmd_fill_rows) over a static corpus of few distinct words against the oracle,
slice by slice.  It found that a vocabulary mass computed as cnt * (1 / len) instead of cnt / len breaks exact ties between a mass and
a capacity (1/17 against 3/51), which upstream's re-charged last shipment (wmd.h:373-375) turns into differences of up to 7e-3
(DESIGN.md section 7).  Run on a GPU box from the repo root: python tools/probe/dbg_fill.py"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from vectorian_amd import core as hip, synth
from oracle import vk_oracle as oracle
from helpers import hip_static_corpus
hip.init(0)
for len_t in (17, 24):
    rng = np.random.default_rng(7 + len_t)
    V, d = 60, 64
    corpus = synth.make_static_corpus(300, 1, 64, V, d, seed=5)
    c, Eb = hip_static_corpus(hip, corpus)
    q_ids = rng.integers(0, V, size=len_t).astype(np.int32)
    off, ids = corpus["sent_off"], corpus["tok_id"]
    for flags in ((False, True, True), (False, False, True), (False, False, False)):
        ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, Q=Eb[q_ids], q_ids=q_ids, algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=9, min_score=-1.0, want_all_scores=True)
        got = c.query(Eb[q_ids], q_token_ids=q_ids, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=9, min_score=-1.0)
        a = c.last_scores(); b = ref["all_scores"]
        bad = np.nonzero(np.abs(a - b) > 1e-5)[0]
        print(len_t, flags, "bad", [(int(s), int(off[s+1]-off[s]), float(a[s]), float(b[s])) for s in bad[:8]])
        for s in bad[:2]:
            sl = ids[off[s]:off[s+1]]
            print("   slice ids", sl.tolist(), "uniq", len(set(sl.tolist())), "q uniq", len(set(q_ids.tolist())), "even idx" , s % 2)
    c.close()
print("---- alignment and injective RWMD on the same corpora")
for len_t in (17, 24, 32):
    rng = np.random.default_rng(7 + len_t)
    V, d = 60, 64
    corpus = synth.make_static_corpus(300, 1, 64, V, d, seed=5)
    c, Eb = hip_static_corpus(hip, corpus)
    q_ids = rng.integers(0, V, size=len_t).astype(np.int32)
    off, ids = corpus["sent_off"], corpus["tok_id"]
    for name, kw in (("linear", dict(gap_s=0.1, gap_t=0.1)), ("rwmd_inj", dict(algorithm=1, rwmd=(True, True, True)))):
        ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, Q=Eb[q_ids], q_ids=q_ids, max_matches=9, min_score=-1.0, want_all_scores=True, **kw)
        got = c.query(Eb[q_ids], q_token_ids=q_ids, q_normalize=False, max_matches=9, min_score=-1.0, **kw)
        a = c.last_scores(); b = ref["all_scores"]
        bad = np.nonzero(np.abs(a - b) > 1e-5)[0]
        print(len_t, name, "bad", [(int(s), int(off[s+1]-off[s]), float(a[s]), float(b[s])) for s in bad[:8]])
    c.close()
