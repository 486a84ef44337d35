"""probe: similarity rows of a batch's winners (vk_query_batch, GEMM path) against those of single queries"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vectorian_amd import core, synth
core.init(0)
rng = np.random.default_rng(0)
n, d = 600, int(sys.argv[1]) if len(sys.argv) > 1 else 64
X = rng.standard_normal((n * 32, d)).astype(np.float32)
c = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=n * 32, n_sentences=n)
c.append_vectors(X, normalize=True)
c.set_sentences(np.arange(0, n * 32 + 1, 32, dtype=np.int64))
c.finalize()
qs = [X[32 * i + 2:32 * i + 2 + 3 + i % 8] + 0.1 * rng.standard_normal((3 + i % 8, d)).astype(np.float32) for i in range(20)]
outs = c.query_batch(qs, algorithm=core.VK_ALG_RWMD, rwmd=(True, True, True), q_normalize=True, max_matches=4, want_flow=True)
for i in (0, 5, 19):
	one = c.query(qs[i], algorithm=core.VK_ALG_RWMD, rwmd=(True, True, True), q_normalize=True, max_matches=4, want_flow=True)
	b = outs[i]
	print(i, "sentences", b.sentence[:b.n], one.sentence[:one.n], "rows none?", b.sim_rows is None,
		"batch rows absmax", None if b.sim_rows is None else float(np.abs(b.sim_rows).max()), "single absmax", float(np.abs(one.sim_rows).max()),
		"max diff", None if b.sim_rows is None else float(np.abs(b.sim_rows[:one.n, :, :one.sim_rows.shape[2]] - one.sim_rows[:one.n]).max()))
