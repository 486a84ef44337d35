"""probe: host and device time of one vk_query_batch call (256 relaxed-WMD queries x 500 k sentences), with and without the winners' similarity rows"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vectorian_amd import core, synth
core.init(0)
n, d = 500000, 300
rng = np.random.default_rng(1)
E = synth.make_vocab(50000, d)
ids = synth.zipf_ids(n * 32, 50000, rng)
c = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=n * 32, n_sentences=n, keep_magnitudes=True)
for a in range(0, n * 32, 1 << 20):
	b = min(n * 32, a + (1 << 20))
	c.append_vectors(E[ids[a:b]] + 0.1 * rng.standard_normal((b - a, d)).astype(np.float32), normalize=True)
c.set_sentences(np.arange(n + 1, dtype=np.int64) * 32)
c.finalize()
qs = [np.ascontiguousarray(E[rng.integers(0, 50000, size=10)], dtype=np.float32) for _ in range(256)]
opts = dict(algorithm=core.VK_ALG_RWMD, rwmd=(True, True, True), q_normalize=True, max_matches=10, min_score=0.0)
for flow in (True, False):
	for rep in range(3):
		t0 = time.perf_counter()
		tops = c.query_batch(qs, want_flow=flow, **opts)
		el = time.perf_counter() - t0
		print("want_flow", flow, "query_batch ms", round(el * 1e3, 2), c.last_timings())
