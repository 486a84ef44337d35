#!/usr/bin/env python3
"""Is the launch heuristic of the scoring kernels (workgroups per CU, launch_sized in vk_score.hip.h) sane away from the
three shapes it was measured on?  For each embedding width d: one contextual corpus of ~`--gbytes` of bf16 rows, queries
of 10 tokens, and the scoring kernel's time with the library's own choice and with VK_BLOCKS_PER_CU forced to 1..5 (the
library caps it at the real occupancy).  Prints one JSON line per (d, gap) with the streaming rate of every setting.

  python tools/sweep_dims.py --dims 64,128,256,300,384,512,768,1024 > gpurun_out/sweep_dims.jsonl
"""

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--dims", default="64,128,256,300,384,512,768,1024")
	ap.add_argument("--gbytes", type=float, default=8.0)
	ap.add_argument("--len-s", type=int, default=32)
	ap.add_argument("--ragged", action="store_true", help="slice lengths U{8..64} instead of --len-s")
	ap.add_argument("--len-t", type=int, default=10, help="query tokens (1 with --len-s 1: the span-embedding index, vk_span_kernel)")
	ap.add_argument("--reps", type=int, default=5)
	ap.add_argument("--max-setting", type=int, default=5)
	ap.add_argument("--precision", choices=["bf16", "f32"], default="bf16", help="how the unit rows are kept in HBM")
	args = ap.parse_args()

	import torch
	from vectorian_amd import core, synth
	core.init(0)
	device = torch.device("cuda", 0)
	V = 20000
	w = (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32)
	gaps = {"linear": 0.1, "exp5": ("table", w)}
	for d in [int(x) for x in args.dims.split(",")]:
		rng = np.random.default_rng(d)
		mean_len = 36 if args.ragged else args.len_s
		esz = 4 if args.precision == "f32" else 2
		n_sent = int(args.gbytes * 1e9 / (mean_len * d * esz))
		lens = rng.integers(8, 65, size=n_sent) if args.ragged else np.full(n_sent, args.len_s)
		off = np.zeros(n_sent + 1, dtype=np.int64)
		np.cumsum(lens, out=off[1:])
		n_tok = int(off[-1])
		E = synth.make_vocab(V, d)
		E_dev = torch.from_numpy(E).to(device)
		corpus = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=n_tok, n_sentences=n_sent, precision=args.precision)
		gen = torch.Generator(device=device)
		gen.manual_seed(7)
		chunk = 1 << 20
		for a in range(0, n_tok, chunk):
			b = min(a + chunk, n_tok)
			idx = torch.randint(0, V, (b - a,), device=device, generator=gen)
			x = (E_dev[idx] + 0.3 * torch.randn((b - a, d), device=device, generator=gen)).contiguous()
			torch.cuda.synchronize()
			corpus.append_vectors_device(x.data_ptr(), b - a, core.VK_F32, normalize=True)
			del x, idx
		corpus.set_sentences(off)
		corpus.finalize()
		qs = [np.ascontiguousarray(E[rng.integers(0, V, size=args.len_t)] + 0.05 * rng.standard_normal((args.len_t, d)).astype(np.float32)) for _ in range(args.reps + 1)]
		for gname, gap in gaps.items():
			row = {"d": d, "gap": gname, "sentences": n_sent, "len_s": "U{8..64}" if args.ragged else args.len_s, "precision": args.precision, "bytes": n_tok * d * esz, "GBps": {}, "ms": {}}
			for setting in ["default"] + [str(i) for i in range(1, args.max_setting + 1)]:
				if setting == "default":
					os.environ.pop("VK_BLOCKS_PER_CU", None)
				else:
					os.environ["VK_BLOCKS_PER_CU"] = setting
				ms = []
				for i, q in enumerate(qs):
					corpus.query(q, algorithm=core.VK_ALG_ALIGN, locality=0, gap_s=gap, gap_t=gap, q_normalize=True, max_matches=10, min_score=0.0, want_flow=True)
					if i:
						ms.append(corpus.last_timings()["score_ms"])
				t = float(np.median(ms))
				row["ms"][setting] = round(t, 4)
				row["GBps"][setting] = round(n_tok * d * esz / (t * 1e-3) / 1e9, 1)
			os.environ.pop("VK_BLOCKS_PER_CU", None)
			best = max((k for k in row["GBps"] if k != "default"), key=lambda k: row["GBps"][k])
			row["best_forced"] = best
			row["default_vs_best"] = round(row["GBps"]["default"] / row["GBps"][best], 4)
			print(json.dumps(row), flush=True)
		corpus.close()
		del E_dev
		torch.cuda.empty_cache()


if __name__ == "__main__":
	main()
