#!/usr/bin/env python3
"""Whole documents as slices (session.partition("document")): time of one query over `--docs` documents of `--min-len` .. `--max-len`
tokens at the C-ABI (vk_query: scoring pass, selection, tracebacks of the winners), per strategy, with the CPU oracle timed beside
it on the same corpus (bounded by --oracle-docs).  One JSON line per strategy.

  python tools/bench_documents.py --docs 2000 --min-len 500 --max-len 5000 --d 300
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--docs", type=int, default=2000)
	ap.add_argument("--min-len", type=int, default=500)
	ap.add_argument("--max-len", type=int, default=5000)
	ap.add_argument("--d", type=int, default=300)
	ap.add_argument("--len-t", type=int, default=10)
	ap.add_argument("--reps", type=int, default=5)
	ap.add_argument("--oracle-docs", type=int, default=64, help="documents the CPU oracle scores (its time is scaled to the corpus by tokens)")
	ap.add_argument("--strategies", default="linear,affine,wsb,rwmd")
	ap.add_argument("--short", type=int, default=0, help="this many 32-token slices after the documents (a corpus of sentences that holds a few documents: the sentences keep their fused kernels)")
	args = ap.parse_args()

	from vectorian_amd import core, synth
	core.init(0)
	rng = np.random.default_rng(2345)
	lens = rng.integers(args.min_len, args.max_len + 1, size=args.docs)
	if args.short > 0:
		lens = np.concatenate((lens, np.full(args.short, 32, dtype=lens.dtype)))
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	T, d = int(off[-1]), args.d
	V = 50000
	E = synth.make_vocab(V, d)
	Xb = np.empty((T, d), dtype=np.uint16)
	for a in range(0, T, 1 << 18):   # in chunks: the fp32 rows of a whole corpus need not exist at once
		b = min(T, a + (1 << 18))
		X = E[synth.zipf_ids(b - a, V, rng)] + 0.1 * rng.standard_normal((b - a, d)).astype(np.float32)
		Xb[a:b] = synth.to_bf16_bits(synth.normalize_rows(X))
	c = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=T, n_sentences=len(lens))
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	# planted query: tokens of one document, spread over 3 x len_t positions
	s = int(rng.integers(0, args.docs))
	a0 = int(off[s]) + int(rng.integers(0, lens[s] - 3 * args.len_t))
	idx = np.sort(rng.choice(np.arange(a0, a0 + 3 * args.len_t), size=args.len_t, replace=False))
	Q = synth.bf16_bits_to_f32(Xb[idx]) + 0.05 * rng.standard_normal((args.len_t, d)).astype(np.float32)
	Qb = synth.to_bf16_bits(synth.normalize_rows(Q))
	w = ("table", (1 - 2.0 ** (-np.arange(0, int(lens.max()) + 1) / 5)).astype(np.float32))
	strategies = {
		"linear": dict(locality=0, gap_s=0.1, gap_t=0.1),
		"affine": dict(locality=0, gap_s=("affine", 0.2, 0.05), gap_t=("affine", 0.2, 0.05)),
		"wsb": dict(locality=0, gap_s=w, gap_t=w),
		"rwmd": dict(algorithm=core.VK_ALG_RWMD, rwmd=(True, True, True)),
	}
	n_o = min(args.oracle_docs, len(lens))
	for name in args.strategies.split(","):
		kw = strategies[name]
		got = c.query(Qb, q_normalize=False, max_matches=10, **kw)   # warm-up (scratch, gap table)
		times, score_ms, flow_ms = [], [], []
		for _ in range(args.reps):
			t0 = time.perf_counter()
			got = c.query(Qb, q_normalize=False, max_matches=10, **kw)
			times.append((time.perf_counter() - t0) * 1e3)
			tm = c.last_timings()
			score_ms.append(tm["score_ms"]); flow_ms.append(tm["flow_ms"])
		line = {"strategy": name, "docs": args.docs, "short_slices": args.short, "tokens": T, "len_min": args.min_len, "len_max": args.max_len, "d": d, "len_t": args.len_t,
			"query_ms": round(float(np.median(times)), 3), "score_kernel_ms": round(float(np.median(score_ms)), 3), "flow_ms": round(float(np.median(flow_ms)), 3),
			"docs_per_s": round(args.docs / (np.median(times) * 1e-3), 1), "tokens_per_s": round(T / (np.median(times) * 1e-3), 1),
			"hbm_frac_of_8TBps": round(T * d * 2 / (np.median(score_ms) * 1e-3) / 8e12, 4), "top": [int(got.sentence[0]), float(got.score[0])], "planted": s}
		if n_o > 0:
			from oracle import vk_oracle as vo
			okw = dict(kw)
			if "algorithm" in okw:
				okw["algorithm"] = vo.ALG_RWMD
			To = int(off[n_o])
			threads = os.cpu_count() or 1
			t0 = time.perf_counter()
			vo.find(layout=vo.LAYOUT_CONTEXTUAL, d=d, sent_off=off[:n_o + 1], X=Xb[:To], Q=Qb, max_matches=10, n_threads=threads, **okw)
			dt = time.perf_counter() - t0
			line["cpu_oracle"] = {"docs": n_o, "tokens": To, "threads": threads, "seconds": round(dt, 3), "scaled_to_corpus_ms": round(dt * T / To * 1e3, 1)}
		print(json.dumps(line), flush=True)
	c.close()


if __name__ == "__main__":
	main()
