"""where one Index.find() spends its time (static layout, 1 M sentences): cProfile of 30 calls + the backend's own phase timings.
  python tools/probe/find_latency.py [--sentences N] [--contextual]"""
import argparse, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import index_throughput as T

ap = argparse.ArgumentParser()
ap.add_argument("--sentences", type=int, default=1000000)
ap.add_argument("--contextual", action="store_true")
args = ap.parse_args()
index, texts, _ = T.build(args.sentences, args.contextual, "local")
for t in texts[:5]:
	index.find(t, n=10)
t0 = time.time()
for t in texts[:30]:
	index.find(t, n=10)
print("ms per find:", (time.time() - t0) / 30 * 1e3, "backend phases of the last call:", index._corpus.last_timings())
pr = cProfile.Profile(); pr.enable()
for t in texts[:30]:
	index.find(t, n=10)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
