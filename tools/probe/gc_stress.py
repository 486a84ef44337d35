"""find_many on three handles from worker threads while another thread collects garbage without pause: a crash here would be a
buffer some C call still uses after Python has let go of it.  python tools/probe/gc_stress.py [seconds]"""
import gc, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_index_sweep as T
from vectorian_amd import core
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
core.init(0)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
stop = [False]

def collector():
	n = 0
	while not stop[0]:
		junk = []
		for _ in range(50):
			a = {"x": np.zeros(17)}; a["self"] = a; junk.append(a)   # cyclic garbage with numpy arrays inside
		del junk
		gc.collect()
		n += 1
	print("collections:", n, flush=True)

th = threading.Thread(target=collector); th.start()
t_end = time.time() + seconds
sessions = queries = 0
seed = 0
while time.time() < t_end:
	rng = np.random.default_rng(424242 + seed); seed += 1
	session, emb, nlp, words = T.build_session(rng)
	strategy, is_align = T.random_strategy(rng)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy)
	part = session.partition("sentence", int(rng.integers(1, 3)), 1)
	index = part.index(sim, nlp=nlp)
	texts = [" ".join(words[int(i)] for i in rng.integers(0, len(words), size=int(rng.integers(1, 14)))) for _ in range(24)]
	for rep in range(3):
		index.find_many(texts, n=5, min_score=-100.0, in_flight=3)
		queries += len(texts)
	index.close()
	sessions += 1
stop[0] = True; th.join()
print("sessions", sessions, "queries", queries, "no crash", flush=True)
