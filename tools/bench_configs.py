#!/usr/bin/env python3
"""Timing of the other BASELINE.json configurations / algorithms on one MI355X (not the
driver's bench: that is bench.py).  Prints one JSON line per run.

  python tools/bench_configs.py --alg rwmd --sentences 1000000
  python tools/bench_configs.py --alg wrd --d 768 --min-len 8 --max-len 64 --sentences 500000
  python tools/bench_configs.py --alg align --gap exp5 --locality global --d 768 --min-len 8 --max-len 64
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
	ap.add_argument("--alg", choices=["align", "rwmd", "wrd"], default="align")
	ap.add_argument("--gap", choices=["exp5", "linear", "affine", "lintable", "convex"], default="exp5", help="lintable: the linear cost 0.1 k handed over as a table (not strictly subadditive); convex: 0.02 k^2 capped at 1 (superadditive at short gaps)")
	ap.add_argument("--locality", choices=["local", "global", "semiglobal"], default="local")
	ap.add_argument("--d", type=int, default=300)
	ap.add_argument("--len-t", type=int, default=10)
	ap.add_argument("--min-len", type=int, default=32)
	ap.add_argument("--max-len", type=int, default=32)
	ap.add_argument("--sentences", type=int, default=1000000)
	ap.add_argument("--steps", type=int, default=10)
	ap.add_argument("--warmup", type=int, default=2)
	ap.add_argument("--k", type=int, default=10)
	ap.add_argument("--layout", choices=["contextual", "static"], default="contextual")
	ap.add_argument("--batch", type=int, default=0, help="queries per vk_query_batch call (config 4: 256)")
	ap.add_argument("--precision", choices=["bf16", "f32"], default="bf16", help="how the unit rows are kept in HBM")
	ap.add_argument("--handles", type=int, default=1, help="queries in flight (vk_corpus_view handles, one host thread each), as bench.py keeps them")
	ap.add_argument("--filter", type=float, default=0.0, help="share of tokens a pos_filter drops: times vk_corpus_filter, then queries the filtered corpus")
	args = ap.parse_args()

	import torch
	from vectorian_amd import core, synth
	core.init(0)
	device = torch.device("cuda", 0)
	V = 50000
	E = synth.make_vocab(V, args.d)
	rng = np.random.default_rng(1)
	lens = rng.integers(args.min_len, args.max_len + 1, size=args.sentences) if args.max_len > args.min_len else np.full(args.sentences, args.min_len)
	off = np.zeros(args.sentences + 1, dtype=np.int64)
	np.cumsum(lens, out=off[1:])
	n_tok = int(off[-1])
	ids = synth.zipf_ids(n_tok, V, rng)
	if args.layout == "static":
		# the reference's static-embedding layout: token ids + vocabulary table, per-query [V x |q|] table
		corpus = core.Corpus(layout=core.VK_LAYOUT_STATIC, d=args.d, n_tokens=n_tok, n_sentences=args.sentences, vocab_size=V)
		corpus.append_vectors(E, normalize=True)
		corpus.set_token_ids(ids)
		corpus.set_sentences(off)
		corpus.finalize()
		w = (1 - 2.0 ** (-np.arange(0, max(65, args.len_t + 1)) / 5)).astype(np.float32)
		gap = {"exp5": ("table", w), "linear": 0.1, "affine": ("affine", 0.2, 0.05)}[args.gap]
		loc = {"local": 0, "global": 1, "semiglobal": 2}[args.locality]
		alg = {"align": core.VK_ALG_ALIGN, "rwmd": core.VK_ALG_RWMD, "wrd": core.VK_ALG_WRD}[args.alg]
		qs = []
		for i in range(args.steps + args.warmup):
			s_ = int(rng.integers(0, args.sentences)); st_ = int(off[s_])
			qi = ids[st_:st_ + args.len_t] if i % 2 == 0 else rng.integers(0, V, size=args.len_t)
			if len(qi) < args.len_t:
				qi = np.concatenate([qi, rng.integers(0, V, size=args.len_t - len(qi))])
			qs.append((np.ascontiguousarray(E[qi], dtype=np.float32), np.asarray(qi, dtype=np.int32)))
		for i in range(args.warmup):
			corpus.query(qs[i][0], q_token_ids=qs[i][1], algorithm=alg, locality=loc, gap_s=gap, gap_t=gap, max_matches=args.k, want_flow=args.alg == "align")
		torch.cuda.synchronize()
		ph = []
		t0 = time.perf_counter()
		for i in range(args.steps):
			q = qs[args.warmup + i]
			corpus.query(q[0], q_token_ids=q[1], algorithm=alg, locality=loc, gap_s=gap, gap_t=gap, max_matches=args.k, want_flow=args.alg == "align")
			ph.append(corpus.last_timings())
		el = time.perf_counter() - t0
		score_ms = float(np.mean([p["score_ms"] for p in ph]))
		cells = float(n_tok) * args.len_t
		print(json.dumps({"layout": "static", "alg": args.alg, "gap": args.gap, "d": args.d, "len_t": args.len_t, "len_s": [args.min_len, args.max_len],
			"sentences": args.sentences, "pairs_per_s": args.sentences * args.steps / el, "ms_per_query": el / args.steps * 1e3,
			"score_kernel_ms": score_ms, "GCUPS": cells / (score_ms * 1e-3) / 1e9,
			"token_id_GBps": n_tok * 4 / (score_ms * 1e-3) / 1e9,
			"phases_ms_mean": {k: float(np.mean([p[k] for p in ph])) for k in ph[0]}}))
		corpus.close()
		return
	corpus = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=args.d, n_tokens=n_tok, n_sentences=args.sentences,
		keep_magnitudes=args.alg == "wrd", precision=args.precision)
	E_dev = torch.from_numpy(E).to(device)
	gen = torch.Generator(device=device)
	gen.manual_seed(7)
	chunk = 1 << 19
	for a in range(0, n_tok, chunk):
		b = min(a + chunk, n_tok)
		idx = torch.from_numpy(ids[a:b].astype(np.int64)).to(device)
		x = E_dev[idx] + 0.3 * torch.randn((b - a, args.d), device=device, generator=gen)
		if args.alg == "wrd":
			x = x * torch.exp(0.25 * torch.randn((b - a, 1), device=device, generator=gen))
		x = x.contiguous()
		torch.cuda.synchronize()
		corpus.append_vectors_device(x.data_ptr(), b - a, core.VK_F32, normalize=True)
		del x, idx
	corpus.set_sentences(off)
	corpus.finalize()

	filter_ms = None
	n_tok_scored = n_tok
	if args.filter > 0.0:
		source = corpus
		drop = (rng.random(n_tok) < args.filter).astype(np.int8)
		source.set_token_pos(drop)
		n_tok_scored = int(n_tok - drop.sum())   # the kernel streams the tokens that pass the filter: the roofline counts those
		source.filtered(pos_mask=2).close()   # warm-up (first use of the scan)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		corpus = source.filtered(pos_mask=2)
		filter_ms = (time.perf_counter() - t0) * 1e3

	w = (1 - 2.0 ** (-np.arange(0, max(65, args.len_t + 1)) / 5)).astype(np.float32)
	gap = {"exp5": ("table", w), "linear": 0.1, "affine": ("affine", 0.2, 0.05),
		"lintable": ("table", (0.1 * np.arange(0, 65)).astype(np.float32)),
		"convex": ("table", np.minimum(0.02 * np.arange(0, 65) ** 2, 1.0).astype(np.float32))}[args.gap]
	loc = {"local": 0, "global": 1, "semiglobal": 2}[args.locality]
	alg = {"align": core.VK_ALG_ALIGN, "rwmd": core.VK_ALG_RWMD, "wrd": core.VK_ALG_WRD}[args.alg]
	qs = []
	for i in range(args.steps + args.warmup):
		if i % 2 == 0:
			s = int(rng.integers(0, args.sentences))
			st = int(off[s])
			qi = ids[st:st + args.len_t]
			if len(qi) < args.len_t:
				qi = np.concatenate([qi, rng.integers(0, V, size=args.len_t - len(qi))])
		else:
			qi = rng.integers(0, V, size=args.len_t)
		qs.append(np.ascontiguousarray(E[qi] + 0.05 * rng.standard_normal((args.len_t, args.d)).astype(np.float32), dtype=np.float32))

	def step(q):
		return corpus.query(q, algorithm=alg, locality=loc, gap_s=gap, gap_t=gap, q_normalize=True, max_matches=args.k,
			min_score=0.0 if loc != 1 else -1e9, want_flow=args.alg == "align")

	if args.batch > 0:
		# BASELINE config 4: a batch of queries per call
		bq = [np.ascontiguousarray(E[rng.integers(0, V, size=args.len_t)] + 0.05 * rng.standard_normal((args.len_t, args.d)).astype(np.float32), dtype=np.float32)
			for _ in range(args.batch)]
		for i in range(0, len(bq), 2):    # half of them planted
			s_ = int(rng.integers(0, args.sentences)); st_ = int(off[s_])
			qi = ids[st_:st_ + args.len_t]
			if len(qi) == args.len_t:
				bq[i] = np.ascontiguousarray(E[qi] + 0.05 * rng.standard_normal((args.len_t, args.d)).astype(np.float32), dtype=np.float32)
		def bstep():
			return corpus.query_batch(bq, algorithm=alg, locality=loc, gap_s=gap, gap_t=gap, q_normalize=True, max_matches=args.k,
				min_score=0.0 if loc != 1 else -1e9, want_flow=False)
		for i in range(args.warmup):
			bstep()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		ph = []
		for i in range(args.steps):
			outs = bstep()
			ph.append(corpus.last_timings())
		el = time.perf_counter() - t0
		score_ms = float(np.mean([p["score_ms"] for p in ph]))
		flops = 2.0 * n_tok * args.batch * args.len_t * args.d
		if args.min_len == 32 and args.max_len == 32:   # vk_rwmd_batch32_kernel: 3 (len_t <= 10) or 2 queries per 32-row tile, K padded to 16
			qpt = 3 if args.len_t <= 10 else 2
			flops_padded = 2.0 * n_tok * ((args.batch + qpt - 1) // qpt) * 32 * ((args.d + 15) // 16 * 16)
		elif args.max_len > args.min_len and args.d <= 320:   # ragged: sentences padded to 32 / 64 tokens on the 32x32x16 kernels, 16 queries per 5 tiles
			padded = float(np.sum((lens + 31) // 32 * 32))
			flops_padded = 2.0 * padded * ((args.batch + 15) // 16 * 5) * 32 * ((args.d + 15) // 16 * 16)
		else:
			padded = float(np.sum((lens + 15) // 16 * 16))
			flops_padded = 2.0 * padded * args.batch * 16 * ((args.d + 31) // 32 * 32)
		print(json.dumps({
			"alg": args.alg, "batch": args.batch, "d": args.d, "len_t": args.len_t, "len_s": [args.min_len, args.max_len],
			"sentences": args.sentences, "pairs_per_s": args.sentences * args.batch * args.steps / el,
			"ms_per_batch": el / args.steps * 1e3, "gemm_kernel_ms": score_ms,
			"algorithmic_TFLOPs": flops / (score_ms * 1e-3) / 1e12, "issued_mfma_TFLOPs": flops_padded / (score_ms * 1e-3) / 1e12,
			"frac_of_2.5PF_algorithmic": flops / (score_ms * 1e-3) / 2.5e15,
			"phases_ms_mean": {k: float(np.mean([p[k] for p in ph])) for k in ph[0]},
			"top_score_q0": float(outs[0].score[0]) if outs[0].n else None}))
		corpus.close()
		return

	for i in range(args.warmup):
		step(qs[i])
	torch.cuda.synchronize()
	phases = []
	if args.handles > 1:
		# several queries in flight: the selection / exact stage / traceback of one query beside the scoring kernel of the next
		from concurrent.futures import ThreadPoolExecutor
		handles = [corpus] + [corpus.view() for _ in range(args.handles - 1)]

		def run(h, q):
			r = h.query(q, algorithm=alg, locality=loc, gap_s=gap, gap_t=gap, q_normalize=True, max_matches=args.k,
				min_score=0.0 if loc != 1 else -1e9, want_flow=args.alg == "align")
			return r, h.last_timings()
		with ThreadPoolExecutor(max_workers=len(handles)) as pool:
			for h in handles:
				run(h, qs[0])          # one at a time first: a handle's kernel queues behind its peer's from the second query on
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			futs = []
			for i in range(args.steps):
				if len(futs) >= len(handles):
					top, ph = futs.pop(0).result()
					phases.append(ph)
				futs.append(pool.submit(run, handles[i % len(handles)], qs[args.warmup + i]))
			for f in futs:
				top, ph = f.result()
				phases.append(ph)
			el = time.perf_counter() - t0
		for h in handles[1:]:
			h.close()
	else:
		t0 = time.perf_counter()
		for i in range(args.steps):
			top = step(qs[args.warmup + i])
			phases.append(corpus.last_timings())
		el = time.perf_counter() - t0
	score_ms = float(np.mean([p["score_ms"] for p in phases]))
	bytes_alg = n_tok_scored * args.d * 2   # (round 2 counted the unfiltered tokens: "hbm_frac_of_8TBps: 1.044" with --filter)
	print(json.dumps({
		"alg": args.alg, "gap": args.gap, "locality": args.locality, "d": args.d, "len_t": args.len_t,
		"len_s": [args.min_len, args.max_len], "sentences": args.sentences, "tokens": n_tok, "tokens_scored": n_tok_scored,
		"pairs_per_s": args.sentences * args.steps / el, "ms_per_query": el / args.steps * 1e3,
		"score_kernel_ms": score_ms, "score_kernel_GBps": bytes_alg / (score_ms * 1e-3) / 1e9,
		"hbm_frac_of_8TBps": bytes_alg / (score_ms * 1e-3) / 8e12,
		"phases_ms_mean": {k: float(np.mean([p[k] for p in phases])) for k in phases[0]},
		"top_score": float(top.score[0]) if top.n else None, "filter_build_ms": filter_ms, "handles": args.handles}))
	corpus.close()


if __name__ == "__main__":
	main()
