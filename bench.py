#!/usr/bin/env python3
"""bench.py -- sentence-alignments/sec of the brute-force alignment search on MI355X.

Headline workload (BASELINE.json configs[1], SURVEY 8d "Config 2"): per GPU 1,000,000 sentences x 32
tokens x 300-d bf16 (contextual layout, one vector per token occurrence), 10-token query,
local alignment, k = 10, min_score = 0.  A "step" is one query against the resident shard:
query upload -> fused similarity (MFMA) + DP kernel -> bounded result set -> flow of the
winners -> results on the host (N > 1: + all-gather of the per-rank result sets + merge).

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5|5wrd|2f32] [--gap exp5|linear]
                  [--locality local|global|semiglobal] [--sentences n] [--no-extra] [--scaling weak|strong]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.  The corpus is sharded by sentence range (weak scaling: every
rank holds its own shard of the workload's per-GPU size); the only exchange is the all-gather of k records.
--scaling strong keeps the configuration's TOTAL fixed instead (config 3: 10 M sentences, config 5: 4 M, configs 2 / 4: 1 M;
--sentences then names the total) and gives every rank total / N of it: the configurations at their literal sizes for every N.

--config picks the workload of the headline `value` (default 2); at N = 1 the other SURVEY 8(d)
configurations are then timed as well, each on its own resident corpus at its per-GPU shape, and reported
in the same JSON line as roofline.by_config = {name: [value, kernel_ms, frac]} (the line stays under 2 KB; the long
form -- workload texts, phases, traffic sources, every entry's full roofline -- goes to stderr and to $VK_BENCH_FULL):
  3     1.25 M x 32 x 300-d per GPU, global alignment (Needleman-Wunsch), linear gap 0.1   (config 3 = 10 M over 8 GPUs)
  4     256 queries x 1 M x 32 x 300-d, relaxed WMD rwmd('nbow'), one GEMM-shaped pass     (MFMA-bound)
  5     1 M sentences of 8..64 tokens x 768-d per GPU, WSB local alignment                  (config 5 = 4 M over 4 GPUs)
  5wrd  the same corpus, Word Rotator's Distance (bound pass + exact EMD of the survivors)
  5rwmd the same corpus, a batch of 256 relaxed-WMD queries (the GEMM-shaped pass over 768-d rows)
  2f32  config 2 with fp32 unit rows (VK_PREC_F32: the reference's own precision, twice the bytes)
  2static  config 2's query over 4 M sentences in the reference's static layout (token ids + per-query table; DP-issue bound)
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOCAB = 50000
K_MATCHES = 10
LEN_T = 10
HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK = 2.5e15    # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PF headline includes 2:1 sparsity)
# VALU issue: 256 CUs x 4 SIMDs, one wave64 vector instruction per 2 cycles and SIMD with two or more waves resident (SIMD-32;
# MI355X_MICROARCH.md cycle constants: `v_fma_f32` (wave64) 2 cyc, one wave alone 4), at the 2.4 GHz maximum clock.  Wave-level
# instructions per second, the unit SQ_INSTS_VALU counts in.
VALU_PEAK = 256 * 4 * 2.4e9 / 2

# the per-GPU shape of each configuration
WORKLOADS = {
	"2": dict(name="config2", total=1000000, n_sent=1000000, min_len=32, max_len=32, d=300, alg="align", locality="local", gap="exp5", prec="bf16"),
	"3": dict(name="config3", total=10000000, n_sent=1250000, min_len=32, max_len=32, d=300, alg="align", locality="global", gap="linear", prec="bf16"),
	"4": dict(name="config4", total=1000000, n_sent=1000000, min_len=32, max_len=32, d=300, alg="rwmd", locality="local", gap="linear", prec="bf16", batch=256),
	"5": dict(name="config5_wsb", total=4000000, n_sent=1000000, min_len=8, max_len=64, d=768, alg="align", locality="local", gap="exp5", prec="bf16",
		noise=0.3, norm_sigma=0.25, magnitudes=True),
	"5wrd": dict(name="config5_wrd", total=4000000, n_sent=1000000, min_len=8, max_len=64, d=768, alg="wrd", locality="local", gap="linear", prec="bf16",
		noise=0.3, norm_sigma=0.25, magnitudes=True),
	# config 5's corpus under a batch of 256 relaxed-WMD queries: the GEMM-shaped pass over 768-d rows (round 3: the 32-row kernels with
	# one fat wave per SIMD; the 16-row kernel of round 2 ran at 0.64 G pairs/s)
	"5rwmd": dict(name="config5_rwmd_batch", total=4000000, n_sent=1000000, min_len=8, max_len=64, d=768, alg="rwmd", locality="local", gap="linear", prec="bf16",
		noise=0.3, norm_sigma=0.25, magnitudes=True, batch=256),
	"2f32": dict(name="config2_f32", n_sent=1000000, min_len=32, max_len=32, d=300, alg="align", locality="local", gap="exp5", prec="f32"),
	# the reference's own layout for static embeddings (fastText / GloVe: token ids + vocabulary table, per-query table [V x |q|],
	# StaticEmbeddingSlice, slice/static.h:71-75): no vectors are streamed, the kernel is bound by the DP's instruction issue
	# several queries with the same options in one call (Index.find_many / vk_query_batch): every token tile is read once per PAIR of
	# queries (vk_score_batch_kernel), so the pairs per second are not bound by |s| d 2 bytes per pair
	"2shared": dict(name="config2_shared_pass", n_sent=1000000, min_len=32, max_len=32, d=300, alg="align", locality="local", gap="exp5", prec="bf16", batch=8, per_pass=2),
	# whole documents as slices (session.partition("document")): one wave per document, its state in global memory
	"docs": dict(name="documents_wsb", n_sent=2000, min_n=2000, min_len=500, max_len=5000, d=300, alg="align", locality="local", gap="exp5", prec="bf16",
		kernel="vk_doc_kernel (one wave per document, skewed sweep, general gaps)", rate_frac=0.03, bound="valu"),
	"docslin": dict(name="documents_linear", n_sent=2000, min_n=2000, min_len=500, max_len=5000, d=300, alg="align", locality="local", gap="linear", prec="bf16",
		kernel="vk_doc_kernel (one wave per document, skewed sweep)", rate_frac=0.5),   # (round 4: within 1.5 x of the HBM roofline -- priced against it)
	# slices of 300 .. 512 tokens (paragraphs, windows of sentences): the same kernel with enough slices to fill the chip
	"mid": dict(name="slices_300_512_linear", n_sent=8000, min_n=8000, min_len=300, max_len=512, d=300, alg="align", locality="local", gap="linear", prec="bf16",
		kernel="vk_doc_kernel (one wave per slice, skewed sweep)", rate_frac=0.6),
	"2static": dict(name="config2_static", n_sent=4000000, min_len=32, max_len=32, d=300, alg="align", locality="local", gap="exp5", prec="bf16", layout="static", bound="valu"),
	# config 4 in the reference's own layout for static embeddings (round 4): ONE similarity table over the vocabulary per batch (an
	# MFMA GEMM 600 x smaller than config 4's) and one gather pass over the token ids with config 4's epilogues -- VALU / L2 bound
	"4static": dict(name="config4_static", n_sent=1000000, min_len=32, max_len=32, d=300, alg="rwmd", locality="local", gap="linear", prec="bf16", batch=256,
		layout="static", bound="valu", kernel="vk_rwmd_static32_kernel (table gather + config 4's epilogues)"),
	# a whole sentence as the query (40 tokens, general gaps): the four-block kernel, VALU-issue bound
	"2q40": dict(name="config2_q40", n_sent=1000000, min_len=32, max_len=32, d=300, alg="align", locality="local", gap="exp5", prec="bf16", len_t=40,
		kernel="vk_score32_kernel (four column blocks)", rate_frac=0.27, bound="valu"),
}
LOCALITIES = {"local": 0, "global": 1, "semiglobal": 2}


def gap_spec(name, max_len=64):
	if name == "linear":
		return 0.1, 0.1, "linear gap u=0.1"
	w = (1 - 2.0 ** (-np.arange(0, max(64, max_len) + 1) / 5)).astype(np.float32)   # smooth_gap_cost(5), up to the longest slice
	return ("table", w), ("table", w), "general gap w(k)=1-2^(-k/5) (Waterman-Smith-Beyer)"


def describe(spec, n_sent):
	"""one short line per workload (the whole JSON line has to stay under 2 KB: the driver keeps a 2,000-character tail)"""
	lens = f"{spec['min_len']}" if spec["min_len"] == spec["max_len"] else f"{spec['min_len']}..{spec['max_len']}"
	rows = f"{n_sent} x {lens} tok x {spec['d']}-d {spec['prec']}" + (f" static layout (ids over {VOCAB} words)" if spec.get("layout") == "static" else " contextual")
	gap = "WSB general gap 1-2^(-k/5)" if spec["gap"] == "exp5" else "linear gap 0.1"
	LEN_T = spec.get("len_t", globals()["LEN_T"])
	if spec["alg"] == "align" and spec.get("batch"):
		return f"{spec['batch']} x {LEN_T}-tok queries/call over {rows} per GPU, {spec['locality']} {gap}, top-{K_MATCHES}+flow, tiles read once per {spec.get('per_pass', 2)} queries"
	if spec["alg"] == "rwmd":
		return f"{spec.get('batch', 1)} x {LEN_T}-tok queries over {rows} per GPU, RWMD rwmd('nbow'), top-{K_MATCHES}"
	if spec["alg"] == "wrd":
		return f"{LEN_T}-tok query over {rows} per GPU, WRD (bound pass + exact EMD), top-{K_MATCHES}"
	return f"{LEN_T}-tok query over {rows} per GPU, {spec['locality']} alignment {gap}, top-{K_MATCHES}+flow"


def build_shard(core, torch, spec, n_sent, rank, device):
	"""synthetic shard generated on the GPU in chunks (clustered vocabulary, Zipf(1.1) token ids, per-token
	noise), handed to the library by device pointer.  Returns the corpus handle, the vocabulary, the token ids (device) and
	the sentence offsets.  `rank` may be a list of ranks: their shards one after the other in ONE corpus (the self-check of the
	sharded path compares the merged result sets with the answer of that corpus)."""
	from vectorian_amd import synth
	ranks = list(rank) if isinstance(rank, (list, tuple)) else [rank]
	d = spec["d"]
	E = synth.make_vocab(VOCAB, d)                                   # seeded, shared by all ranks
	lens = []
	for r in ranks:
		rng = np.random.default_rng(synth.SEED_CORPUS + 7919 * r)
		if spec["max_len"] > spec["min_len"]:
			lens.append(rng.integers(spec["min_len"], spec["max_len"] + 1, size=n_sent))
		else:
			lens.append(np.full(n_sent, spec["min_len"]))
	n_all = n_sent * len(ranks)
	off = np.zeros(n_all + 1, dtype=np.int64)
	np.cumsum(np.concatenate(lens), out=off[1:])
	n_tok = int(off[-1])
	static = spec.get("layout") == "static"
	if static:
		corpus = core.Corpus(layout=core.VK_LAYOUT_STATIC, d=d, n_tokens=n_tok, n_sentences=n_all, vocab_size=VOCAB, precision=spec["prec"])
		corpus.append_vectors(E, normalize=True)
	else:
		corpus = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=n_tok, n_sentences=n_all,
			keep_magnitudes=bool(spec.get("magnitudes")), precision=spec["prec"])
	E_dev = torch.from_numpy(E).to(device)
	p = 1.0 / np.arange(1, VOCAB + 1) ** 1.1
	cdf = np.cumsum(p)
	cdf_dev = torch.from_numpy(cdf / cdf[-1]).to(device)
	ids = torch.empty(n_tok, dtype=torch.int32, device=device)
	chunk = 1 << 20 if d <= 320 else 1 << 19
	noise, sigma = spec.get("noise", 0.1), spec.get("norm_sigma", 0.0)
	for i, r in enumerate(ranks):
		gen = torch.Generator(device=device)
		gen.manual_seed(1000 + r)
		t0, t1 = int(off[i * n_sent]), int(off[(i + 1) * n_sent])   # this rank's tokens: generated as if the shard stood alone
		for a in range(t0, t1, chunk):
			b = min(a + chunk, t1)
			idx = torch.searchsorted(cdf_dev, torch.rand(b - a, device=device, generator=gen, dtype=torch.float64)).clamp_(max=VOCAB - 1)
			ids[a:b] = idx.to(torch.int32)
			if static:
				continue
			x = E_dev[idx] + noise * torch.randn((b - a, d), device=device, generator=gen, dtype=torch.float32)
			if sigma > 0:
				x = x * torch.exp(sigma * torch.randn((b - a, 1), device=device, generator=gen, dtype=torch.float32))
			x = x.contiguous()
			torch.cuda.synchronize()
			corpus.append_vectors_device(x.data_ptr(), b - a, core.VK_F32, normalize=True)
			del x, idx
	if static:
		corpus.set_token_ids(ids.cpu().numpy())
	corpus.set_sentences(off)
	corpus.finalize()
	del E_dev, cdf_dev
	torch.cuda.empty_cache()
	return corpus, E, ids, off


def make_queries(E, ids, off, n_queries, seed, static=False, len_t=LEN_T):
	"""half of the queries are noisy copies of LEN_T consecutive tokens of a corpus sentence (planted hits), half random;
	static layout: the words themselves (their vectors and vocabulary ids)"""
	LEN_T = len_t
	rng = np.random.default_rng(seed)
	n_sent = len(off) - 1
	d = E.shape[1]
	qs = []
	for i in range(n_queries):
		qi = None
		if i % 2 == 0:
			for _ in range(64):
				s = int(rng.integers(0, n_sent))
				ln = int(off[s + 1] - off[s])
				if ln >= LEN_T:
					st = int(off[s]) + int(rng.integers(0, ln - LEN_T + 1))
					qi = ids[st:st + LEN_T].cpu().numpy().astype(np.int64)
					break
		if qi is None:
			qi = rng.integers(0, VOCAB, size=LEN_T)
		if static:
			qs.append((np.ascontiguousarray(E[qi], dtype=np.float32), np.asarray(qi, dtype=np.int32)))
			continue
		qs.append(np.ascontiguousarray(E[qi] + 0.05 * rng.standard_normal((LEN_T, d)).astype(np.float32), dtype=np.float32))
	return qs


def usable_cores():
	"""host cores this process may actually run on: the affinity mask and the cgroup CPU quota, not the machine's count"""
	n = os.cpu_count() or 1
	try:
		n = min(n, len(os.sched_getaffinity(0)))
	except (AttributeError, OSError):
		pass
	for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
		try:
			with open(path) as f:
				parts = f.read().split()
			if path.endswith("cpu.max"):
				if parts[0] != "max":
					n = min(n, max(1, int(round(int(parts[0]) / int(parts[1])))))
			else:
				quota = int(parts[0])
				with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
					period = int(f.read().split()[0])
				if quota > 0:
					n = min(n, max(1, int(round(quota / period))))
			break
		except (OSError, ValueError, IndexError):
			continue
	return max(1, n)


def cpu_baseline(spec, budget_s=10.0):
	"""The CPU restatement of the reference algorithm (oracle/, kind "port": the reference's
	own C++ path cannot be built offline) on a bounded sample of the same workload:
	all host cores, static sentence ranges per thread (the analogue of the reference's
	thread-per-document pool, vectorian/index.py:544-558), threads started once per batch.
	`value` computes every cosine with the port's scalar dot products; `blas_value` is the form closest to the
	reference's contextual path: ONE fp32 sgemm per query for the similarity matrix of the whole sample
	(numpy @ = BLAS, as vectorian/sim/vector.py:66-78 via metric/contextual.cpp:26-63), then the port's DP over its rows."""
	from oracle import vk_oracle as vo
	from vectorian_amd import synth
	cores = usable_cores()
	n = 8192
	d, ls0, ls1 = spec["d"], spec["min_len"], spec["max_len"]
	corpus = synth.make_contextual_corpus(n, ls0, ls1, VOCAB, d, noise=spec.get("noise", 0.1), norm_sigma=spec.get("norm_sigma", 0.0))
	Xn = synth.normalize_rows(corpus["X"])
	Xb = synth.to_bf16_bits(Xn)
	qraw = synth.make_queries(corpus, 16, LEN_T)
	qs = [synth.to_bf16_bits(synth.normalize_rows(q["vectors"])) for q in qraw]
	gs, gt, _ = gap_spec(spec["gap"], spec["max_len"])
	alg = {"align": vo.ALG_ALIGN, "rwmd": vo.ALG_RWMD, "wrd": vo.ALG_WRD}[spec["alg"]]
	kw = dict(layout=vo.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, algorithm=alg, locality=LOCALITIES[spec["locality"]],
		gap_s=gs, gap_t=gt, max_matches=K_MATCHES, min_score=0.0, n_threads=cores)
	if spec["alg"] == "wrd":
		kw["X_mag"] = vo.magnitudes(corpus["X"])
		kw["Q_mags"] = None
	def rate(budget, n_sent, cap, **over):
		k2 = dict(kw, **over)
		mags = (lambda m: [vo.magnitudes(qraw[i % len(qraw)]["vectors"]) for i in range(m)]) if spec["alg"] == "wrd" else (lambda m: None)
		t1 = time.perf_counter()
		vo.find_many(Qs=qs[:2], **dict(k2, Q_mags=mags(2)))          # calibrate the batch size to the budget
		per_q = (time.perf_counter() - t1) / 2
		b = int(max(2, min(cap, budget / max(per_q, 1e-6))))
		t1 = time.perf_counter()
		vo.find_many(Qs=[qs[i % len(qs)] for i in range(b)], **dict(k2, Q_mags=mags(b)))
		el = time.perf_counter() - t1
		return n_sent * b / el, b, el
	value, batch, el = rate(budget_s, n, 4096)
	out = {
		"value": value, "unit": "sentence-alignments/sec", "cores": cores, "kind": "port",
		"sample": f"{batch} queries x {n} sent x {ls0 if ls0 == ls1 else str(ls0) + '..' + str(ls1)} tok x {d}-d, {cores} threads, {el:.1f} s (C restatement; reference not runnable offline)"}
	# beside it: one thread
	out["single_thread_value"] = rate(2.5, n, 1024, n_threads=1)[0]
	if spec["alg"] == "align":
		# BLAS leg: S = X . Q^T by sgemm (fp32 unit rows, as the reference computes it), clip, then the port's DP
		def blas_run(b):
			S = [np.clip(Xn @ synth.normalize_rows(qraw[i % len(qraw)]["vectors"]).T, 0.0, 1.0) for i in range(b)]
			vo.find_many(Qs=[qs[i % len(qs)] for i in range(b)], S_rows=S, **kw)
		t1 = time.perf_counter()
		blas_run(2)
		per_q = (time.perf_counter() - t1) / 2
		b = int(max(2, min(2048, 4.0 / max(per_q, 1e-6))))
		t1 = time.perf_counter()
		blas_run(b)
		out["blas_value"] = n * b / (time.perf_counter() - t1)
		out["blas_note"] = f"one numpy sgemm [{Xn.shape[0]}x{d}]x[{d}x{LEN_T}] per query + the port's DP"
		# the reference's own static layout (token ids + per-query table [V x |q|])
		n_st = 16 * n      # the per-query table over the vocabulary is amortised over the slices, as in the full workload
		st = synth.make_static_corpus(n_st, ls0, ls1, VOCAB, d)
		Eb = synth.to_bf16_bits(synth.normalize_rows(st["E"]))
		out["static_layout_value"] = rate(2.5, n_st, 1024, layout=vo.LAYOUT_STATIC, X=None, sent_off=st["sent_off"], tok_id=st["tok_id"], E=Eb)[0]
	return out


class Runner:
	"""Keeps queries in flight on one resident shard: two handles (vk_corpus_view: shared arrays, own stream and
	workspaces), one host thread each: query i + 1 is scored while the result set of query i is selected, retraced and
	copied out (round 2, general gaps with the query tile in LDS so that the traceback runs beside the next scoring kernel:
	343 M/s with two handles and with three; the register form needed three: 341 against 326 M/s).  Every query is
	complete (top-k with flow on the host; with several ranks: merged across ranks) inside the timed region."""

	def __init__(self, core, torch, corpus, spec, n_sent, dist, xdev, rank, n_handles, gap_name=None, locality=None, batch_handles=None):
		from concurrent.futures import ThreadPoolExecutor
		from vectorian_amd import shards
		self.core, self.torch, self.corpus, self.spec, self.n_sent = core, torch, corpus, spec, n_sent
		self.dist, self.xdev, self.rank, self.shards = dist, xdev, rank, shards
		self.batch = int(spec.get("batch", 0))
		# batches: two handles as well -- the selection, the copies and the host part of one batch run beside the GEMM of the next
		# (the turn passes after the selection, vk_batch.cpp): 40.9 -> 39.1 ms per batch of 256
		n_batch_handles = batch_handles or max(1, int(os.environ.get("VK_BENCH_BATCH_HANDLES", "2")))
		self.handles = [corpus] + [corpus.view() for _ in range((n_batch_handles if self.batch else n_handles) - 1)]
		self.pool = ThreadPoolExecutor(max_workers=len(self.handles))
		self.inflight, self.pending, self.unsent = [], [], []
		self.gather_batch = max(1, int(os.environ.get("VK_BENCH_GATHER_BATCH", "4")))   # result sets of this many queries travel in one all-gather
		self.gather_depth = int(os.environ.get("VK_BENCH_GATHER_DEPTH", "1"))   # a collective gets this many further exchanges to complete before anyone waits for it
		self.submitted = 0
		self.score_ms, self.phases, self.retired_at, self.median_gap_ms, self.gap_min_max_ms = [], [], [], None, None
		gs, gt, _ = gap_spec(gap_name or spec["gap"], spec["max_len"])
		alg = {"align": core.VK_ALG_ALIGN, "rwmd": core.VK_ALG_RWMD, "wrd": core.VK_ALG_WRD}[spec["alg"]]
		self.options = dict(algorithm=alg, locality=LOCALITIES[locality or spec["locality"]], gap_s=gs, gap_t=gt, q_normalize=True,
			max_matches=K_MATCHES, min_score=0.0, want_flow=spec["alg"] == "align")

	def _run(self, h, q):
		if self.batch:
			if isinstance(q[0], tuple):   # static layout: the vectors and the vocabulary ids of every query's words
				tops = h.query_batch([v for v, _ in q], token_ids=[i for _, i in q], **self.options)
			else:
				tops = h.query_batch(q, **self.options)
			return tops, h.last_timings()
		if isinstance(q, tuple):   # static layout: vectors and vocabulary ids of the query's words
			top = h.query(q[0], q_token_ids=q[1], **self.options)
		else:
			top = h.query(q, **self.options)
		return [top], h.last_timings()

	def _exchange(self):
		if self.unsent:
			# per-rank result sets -> all ranks (RCCL all-gather over xGMI), then ResultSet.extend
			self.pending.append(self.shards.allgather_start(list(self.unsent), self.rank * self.n_sent, K_MATCHES, device=self.xdev))
			self.unsent.clear()

	def _drain(self, keep=0):
		while len(self.pending) > keep:
			self.shards.allgather_finish(self.pending.pop(0))   # global top-k of earlier queries on every rank

	def _retire(self):
		tops, ph = self.inflight.pop(0).result()
		self.retired_at.append(time.perf_counter())
		self.score_ms.append(ph["score_ms"])
		self.phases.append(ph)
		if self.dist is None:
			return
		self.unsent.extend(tops)
		if len(self.unsent) >= self.gather_batch:
			self._exchange()
		self._drain(keep=self.gather_depth)

	def step(self, q):
		h = self.handles[self.submitted % len(self.handles)]
		self.submitted += 1
		self.inflight.append(self.pool.submit(self._run, h, q))
		if len(self.inflight) >= len(self.handles):
			self._retire()

	def sync(self):
		while self.inflight:
			self._retire()
		self._exchange()
		self._drain()
		self.torch.cuda.synchronize()
		if self.dist is not None:
			self.dist.barrier()
			self.torch.cuda.synchronize()

	def run(self, queries, warmup, steps):
		"""queries: warmup + steps entries (an entry = one query, or the list of a batch).  Returns the elapsed seconds of the
		timed steps."""
		for i in range(warmup):
			self.step(queries[i])
			if i < len(self.handles):
				self.sync()   # the first query of a handle has no predecessor to queue behind: one at a time, so that no two
				              # scoring kernels share the GPU (afterwards a handle's kernel waits for its peer's on the device)
		self.sync()
		self.score_ms.clear()
		self.phases.clear()
		self.retired_at.clear()
		import gc
		gc.collect()
		gc.disable()     # no collector pause of the host threads inside the timed region (nothing is skipped: it runs right after)
		try:
			t0 = time.perf_counter()
			for i in range(steps):
				self.step(queries[warmup + i])
			self.sync()
			elapsed = time.perf_counter() - t0
		finally:
			gc.enable()
		gaps_ = np.diff(np.array([t0] + self.retired_at)) * 1e3 if self.retired_at else None
		self.median_gap_ms = float(np.median(gaps_)) if gaps_ is not None else None
		# (the first gap holds the pipeline's fill: the spread is taken over the steps after it)
		self.gap_min_max_ms = [float(gaps_[1:].min()), float(gaps_[1:].max())] if gaps_ is not None and len(gaps_) > 1 else None
		if os.environ.get("VK_BENCH_TRACE"):   # where the time between completed queries went (stalls show as single long gaps)
			gaps = np.diff(np.array([t0] + self.retired_at)) * 1e3
			top = np.argsort(-gaps)[:4]
			print(f"[trace] {self.spec['name']}: {steps} steps, median gap {np.median(gaps):.2f} ms, longest " +
				", ".join(f"{gaps[i]:.1f} ms at step {i}" for i in top), file=sys.stderr)
		return elapsed

	def close(self):
		self.pool.shutdown()
		for h in self.handles[1:]:   # views first, then the handle that owns the arrays (the caller closes that one)
			h.close()


def roofline_of(spec, n_sent, n_tok, kern_s):
	"""the dominant kernel against the roofline that bounds it: algorithmic bytes = |s| * d * 2 per (query,
	sentence) pair (bf16 token vectors read once; fp32 rows: * 4; WRD: + 4 bytes of magnitude per token), padding not
	counted; config 4: 2 * |s| * |q| * d flops per pair against the dense bf16 MFMA peak"""
	d = spec["d"]
	LEN_T = spec.get("len_t", globals()["LEN_T"])
	if spec.get("bound") == "valu":
		# DP-bound kernels (the static layout, whole documents, queries of 33..64 tokens): neither HBM nor MFMA binds them, the
		# vector ALU's issue port does.  achieved = wave-level VALU instructions per second: SQ_INSTS_VALU of the dominant kernel
		# from a kept --pmc pass (profiles/valu.json, per token of the launch: the count is linear in the tokens of a fixed shape)
		# x this launch's tokens / the kernel's duration, against VALU_PEAK.  GCUPS and the HBM share are stated beside it.
		nbytes = (4 * n_tok) if spec.get("layout") == "static" else n_tok * d * 2
		per_tok, src = valu_per_token(spec["name"])
		ach = per_tok * n_tok / kern_s if per_tok else None
		return {"bound": "valu", "achieved": (ach / 1e9) if ach else None, "peak": VALU_PEAK / 1e9, "unit": "Ginst/s", "frac": (ach / VALU_PEAK) if ach else None,
			"kernel": spec.get("kernel") or "vk_score_kernel (static layout)", "kernel_ms": kern_s * 1e3,
			"valu_insts_per_launch": per_tok * n_tok if per_tok else None, "valu_source": src,
			"gcups": n_tok * LEN_T * max(1, int(spec.get("batch", 0))) / kern_s / 1e9, "hbm_frac": nbytes / kern_s / HBM_PEAK, "algorithmic_bytes_per_launch": nbytes,
			"note": "VALU-issue bound: wave-level vector instructions/s (SQ_INSTS_VALU) against CUs x 4 SIMDs x 2.4 GHz / 2 cycles"}
	if spec.get("batch") and spec["alg"] == "align":
		# shared pass: kern_s is the sum over the call's passes (ceil(batch / per_pass)); one pass streams the corpus once
		passes = -(-spec["batch"] // spec.get("per_pass", 2))
		nbytes = n_tok * d * 2
		ach = nbytes / (kern_s / passes)
		return {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
			"kernel": "vk_score_batch_kernel", "kernel_ms": kern_s / passes * 1e3, "algorithmic_bytes_per_launch": nbytes,
			"queries_per_launch": spec.get("per_pass", 2),
			"note": "one launch scores the corpus against 2 queries: the DPs of both, not the stream, bound it"}
	if spec.get("batch"):
		flops = 2.0 * n_tok * spec["batch"] * LEN_T * d
		ach = flops / kern_s
		return {"bound": "mfma", "achieved": ach / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK,
			"kernel": "vk_rwmd_batch32d_kernel", "kernel_ms": kern_s * 1e3, "algorithmic_flops_per_launch": flops}
	if spec.get("layout") == "static":
		# neither roofline binds this layout (SURVEY 8d "Mode B"): 4 bytes of token id per token are all that is streamed; the
		# honest figures are cell updates per second and the (small) share of HBM bandwidth
		nbytes = 4 * n_tok
		ach = nbytes / kern_s
		return {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
			"kernel": "vk_score_kernel (static layout)", "kernel_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": nbytes,
			"gcups": n_tok * LEN_T / kern_s / 1e9, "note": "DP-issue bound: token ids + an L2-resident per-query table; GCUPS is the figure of merit"}
	nbytes = n_tok * d * (4 if spec["prec"] == "f32" else 2) + (4 * n_tok if spec["alg"] == "wrd" else 0)
	ach = nbytes / kern_s
	return {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
		"kernel": spec.get("kernel") or ("vk_score_kernel" + (" (WRD bound pass)" if spec["alg"] == "wrd" else "")), "kernel_ms": kern_s * 1e3,
		"algorithmic_bytes_per_launch": nbytes}


def valu_per_token(name):
	"""wave-level VALU instructions per corpus token of the configuration's dominant kernel, from the committed counter pass
	(profiles/valu.json: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU of this script, tools/pmc_valu.sh) -- not measured in this run"""
	path = os.path.join(ROOT, "profiles", "valu.json")
	try:
		e = json.load(open(path)).get(name)
	except Exception:
		e = None
	if not e or not e.get("tokens_per_launch"):
		return None, None
	return e["valu_insts_per_launch"] / e["tokens_per_launch"], "profiles/valu.json (" + e.get("source", "rocprofv3 --pmc SQ_INSTS_VALU") + "); not measured in this run"


def traffic_of(name, gap, n_sent, nominal):
	"""HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json: rocprofv3 --pmc
	FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this script, gfx950 correction applied) -- not measured in this run"""
	path = os.path.join(ROOT, "profiles", "traffic.json")
	if n_sent != nominal or not os.path.exists(path):
		return None, None
	try:
		t = json.load(open(path))
	except Exception:
		return None, None
	e = t.get(name + ":" + gap) or (t.get(gap) if name == "config2" else None)
	if not e:
		return None, None
	return e.get("hbm_bytes_per_launch"), "profiles/traffic.json (" + e.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, round 1") + "); not measured in this run"


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=40)
	ap.add_argument("--warmup", type=int, default=5)
	ap.add_argument("--config", choices=sorted(WORKLOADS), default="2", help="workload of the headline value")
	ap.add_argument("--gap", choices=["exp5", "linear"], default=None, help="gap costs of the headline alignment workload (default: the config's)")
	ap.add_argument("--locality", choices=sorted(LOCALITIES), default=None, help="locality of the headline alignment workload (default: the config's)")
	ap.add_argument("--sentences", type=int, default=0, help="sentences per GPU of the headline workload (default: the config's per-GPU size)")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-pipeline", action="store_true", help="one handle, one query at a time")
	ap.add_argument("--no-extra", action="store_true", help="headline workload only (no \"configs\" object)")
	ap.add_argument("--extra", default="4,4static,3,2f32,2static,2shared,2q40,5,5wrd,5rwmd,docs,docslin,mid", help="the other configurations timed at N = 1 after the headline")
	ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: every rank holds the configuration's per-GPU shard; "
		"strong: the configuration's total (config 3: 10 M, config 5: 4 M sentences) is divided among the ranks")
	ap.add_argument("--extra-steps", type=int, default=12)
	ap.add_argument("--extra-warmup", type=int, default=4)
	ap.add_argument("--extra-min-ms", type=float, default=300.0, help="the extra configurations run at least this long inside their timed region")
	ap.add_argument("--extra-scale", type=float, default=1.0, help="scales the sentence counts of the extra configurations (rehearsals)")
	ap.add_argument("--selfcheck", action="store_true", help="N > 1, small --sentences: after the timed region one fixed query goes through the "
		"sharded path (per-rank result sets, all-gather, merge) and rank 0 compares the merged result set with the answer of ONE corpus "
		"holding all ranks' shards (\"selfcheck\" in the JSON line)")
	args = ap.parse_args()

	import torch
	from vectorian_amd import core

	world = int(os.environ.get("WORLD_SIZE", "1"))
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if world != args.gpus:
		if rank == 0 and world == 1 and args.gpus > 1:
			print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
			sys.exit(2)
	if not torch.cuda.is_available():
		print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
		sys.exit(2)
	# one rank per GPU; VK_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks on one GPU
	backend = os.environ.get("VK_BENCH_BACKEND", "nccl")
	dev_index = local_rank % torch.cuda.device_count()
	torch.cuda.set_device(dev_index)
	device = torch.device("cuda", dev_index)
	core.init(dev_index)

	dist = None
	force_dist = os.environ.get("VK_BENCH_FORCE_DIST") == "1"   # rehearsal: the N > 1 code path (RCCL calls included) with one rank
	if world > 1 or force_dist:
		import torch.distributed as dist
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		if force_dist and world == 1:   # no launcher in the one-rank rehearsal
			for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
				os.environ.setdefault(k_, v_)
		# RCCL writes a version banner to stdout (fd 1) when its communicator comes up; stdout is for the ONE JSON line:
		# fd 1 points at stderr until the first collective has run
		sys.stdout.flush()
		saved_fd1 = os.dup(1)
		os.dup2(2, 1)
		try:
			if backend == "nccl":
				opts = None
				try:   # the records are tiny and latency-bound: let the collective's kernel pass the scoring kernel in the queue
					opts = dist.ProcessGroupNCCL.Options()
					opts.is_high_priority_stream = True
				except Exception:
					opts = None
				dist.init_process_group(backend="nccl", device_id=device, pg_options=opts)
			else:
				dist.init_process_group(backend=backend)
			warm = torch.zeros(1, device=device if backend == "nccl" else "cpu")
			dist.all_reduce(warm)
			if backend == "nccl":
				torch.cuda.synchronize()
		finally:
			sys.stdout.flush()
			os.dup2(saved_fd1, 1)
			os.close(saved_fd1)
	xdev = device if backend == "nccl" else torch.device("cpu")   # where the exchanged records live

	n_handles = 1 if args.no_pipeline else max(1, int(os.environ.get("VK_BENCH_HANDLES", "2")))

	def measure(key, n_sent, warmup, steps, use_dist, gap=None, locality=None, keep=None):
		"""builds (or takes over) the resident shard of a workload, runs warmup + steps, returns the report entry"""
		spec = dict(WORKLOADS[key])
		if gap and spec["alg"] == "align":
			spec["gap"] = gap
		if locality and spec["alg"] == "align":
			spec["locality"] = locality
		shape = (n_sent, spec["min_len"], spec["max_len"], spec["d"], spec["prec"], bool(spec.get("magnitudes")), spec.get("layout"))
		if keep is not None and keep.get("shape") == shape:
			corpus, E, ids, off = keep["shard"]
		else:
			if keep is not None and keep.get("shard"):
				keep["shard"][0].close()
				keep.clear()
				torch.cuda.empty_cache()
			corpus, E, ids, off = build_shard(core, torch, spec, n_sent, rank, device)
			if keep is not None:
				keep["shape"] = shape
				keep["shard"] = (corpus, E, ids, off)
		n_tok = int(off[-1])
		batch = int(spec.get("batch", 0))
		len_t = spec.get("len_t", LEN_T)
		if batch:
			pool_q = make_queries(E, ids, off, batch + 8, seed=3456, static=spec.get("layout") == "static")
			queries = [[pool_q[(i + j) % len(pool_q)] for j in range(batch)] for i in range(warmup + steps)]
		else:
			queries = make_queries(E, ids, off, warmup + steps, seed=3456, static=spec.get("layout") == "static", len_t=len_t)
			if use_dist is not None and spec.get("layout") != "static":   # one query stream for the whole job: rank 0's
				qt = torch.from_numpy(np.stack(queries)).to(xdev)
				use_dist.broadcast(qt, src=0)
				queries = list(qt.cpu().numpy())
		r = Runner(core, torch, corpus, spec, n_sent, use_dist, xdev, rank, n_handles)
		try:
			elapsed = r.run(queries, warmup, steps)
		finally:
			r.close()
		kernel_alone_ms = None
		shared_pass = spec["alg"] == "align" and batch > 0
		if (spec["alg"] == "wrd" and n_handles > 1) or shared_pass:
			# The bound passes of the queries in flight do not take turns (vk_query.cpp: two passes sharing the chip fill each
			# other's epilogue gaps), so a launch's own events span the work of its neighbours too.  The kernel's duration for the
			# roofline is taken one query at a time, in the same process, right after the timed region.
			# (the shared passes of two calls in flight overlap in the same way)
			r1 = Runner(core, torch, corpus, spec, n_sent, None, xdev, rank, 1, batch_handles=1)
			try:
				r1.run(queries, 1, min(4, steps))
			finally:
				r1.close()
			kernel_alone_ms = float(np.mean(r1.score_ms))
		if keep is None:
			corpus.close()
		if use_dist is not None:
			t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
			use_dist.all_reduce(t, op=use_dist.ReduceOp.MAX)
			elapsed = float(t.item())
		pairs = n_sent * max(1, batch) * (world if use_dist is not None else 1) * steps
		kern_s = (kernel_alone_ms if kernel_alone_ms is not None else float(np.mean(r.score_ms))) * 1e-3
		entry = {
			"workload": describe(spec, n_sent), "value": pairs / elapsed, "unit": "sentence-alignments/sec",
			"steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "dtype": spec["prec"],
			"ms_per_step_median": r.median_gap_ms,   # median time between completed steps: a host stall shows in ms_per_step, not here
			"step_gap_ms": r.gap_min_max_ms,         # shortest and longest time between two completed steps of the timed region
			"sentences_per_gpu": n_sent, "tokens_per_gpu": n_tok, "d": spec["d"],
			"roofline": roofline_of(spec, n_sent, n_tok, kern_s),
			"phases_ms_mean": {k: float(np.mean([p[k] for p in r.phases])) for k in r.phases[0]} if r.phases else {},
		}
		entry["kernel"], entry["kernel_ms"] = entry["roofline"]["kernel"], entry["roofline"]["kernel_ms"]
		if kernel_alone_ms is not None:
			entry["roofline"]["kernel_ms_note"] = ("one call at a time (4 after the timed region); with several in flight the passes "
				f"overlap on the chip and a call's own events span {float(np.mean(r.score_ms)):.2f} ms")
		return spec, entry

	keep = {}
	head = WORKLOADS[args.config]
	if args.scaling == "strong":
		# the configuration's total, divided: rank r holds sentences [r, r + 1) * (total // world) of the job's corpus
		total = args.sentences or head.get("total", head["n_sent"])
		n_sent = max(1, total // world)
	else:
		n_sent = args.sentences or head["n_sent"]
	spec, entry = measure(args.config, n_sent, args.warmup, args.steps, dist, gap=args.gap, locality=args.locality, keep=keep)
	traffic, traffic_source = traffic_of(spec["name"], spec["gap"], n_sent, head["n_sent"])
	if entry["roofline"]["bound"] != "hbm":
		traffic, traffic_source = None, None

	# ---- self-check of the sharded path (tests/test_gpu_bench_ranks.py runs this with two gloo ranks on one GPU)
	selfcheck = None
	if args.selfcheck and dist is not None and spec["alg"] == "align" and spec.get("layout") != "static" and not spec.get("batch"):
		from vectorian_amd import shards as _shards
		corpus = keep["shard"][0]
		E = keep["shard"][1]
		rq = np.random.default_rng(777)
		qv = np.ascontiguousarray(E[rq.integers(0, VOCAB, size=LEN_T)] + 0.05 * rq.standard_normal((LEN_T, spec["d"])).astype(np.float32), dtype=np.float32)
		gs_, gt_, _ = gap_spec(spec["gap"], spec["max_len"])
		opts = dict(algorithm=core.VK_ALG_ALIGN, locality=LOCALITIES[spec["locality"]], gap_s=gs_, gap_t=gt_, q_normalize=True,
			max_matches=K_MATCHES, min_score=0.0 if spec["locality"] == "local" else -1e9, want_flow=True)
		local = corpus.query(qv, **opts)
		merged = _shards.allgather_finish(_shards.allgather_start([local], rank * n_sent, K_MATCHES, device=xdev))[0]
		if rank == 0:
			whole = build_shard(core, torch, spec, n_sent, list(range(world)), device)[0]   # all ranks' shards, one corpus, no exchange
			one = whole.query(qv, **opts)
			whole.close()
			n_ = int(one.n)
			same = (int(merged.n) == n_ and (merged.sentence[:n_] == one.sentence[:n_]).all()
				and (merged.score[:n_].view(np.uint32) == one.score[:n_].view(np.uint32)).all() and (merged.mapping[:n_] == one.mapping[:n_]).all())
			selfcheck = {"ok": bool(same), "n": n_, "merged": [int(x) for x in merged.sentence[:int(merged.n)]], "one_corpus": [int(x) for x in one.sentence[:n_]],
				"ranks_with_winners": sorted({int(x) // n_sent for x in merged.sentence[:int(merged.n)]})}
		dist.barrier()

	# what the collective backend itself saw (the first real multi-GPU run must prove that RCCL ran with N ranks on N devices)
	ranks = {"launched": world, "backend": backend if dist is not None else None, "world_seen_by_backend": 1, "devices": [dev_index]}
	if dist is not None:
		ranks["world_seen_by_backend"] = int(dist.get_world_size())
		seen = [None] * dist.get_world_size()
		dist.all_gather_object(seen, (rank, dev_index, os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")))
		ranks["devices"] = [d_ for _, d_, _ in sorted(seen)]
		vis = sorted({v_ for _, _, v_ in seen if v_})
		if vis:
			ranks["visible"] = vis

	def compact_roofline(e):
		r = e["roofline"]
		return {"bound": r["bound"], "achieved": None if r["achieved"] is None else round(r["achieved"], 1), "peak": r["peak"], "unit": r["unit"],
			"frac": None if r["frac"] is None else round(r["frac"], 4), "kernel": r["kernel"], "kernel_ms": round(r["kernel_ms"], 4)}

	out = full = None
	if rank == 0:
		roof = compact_roofline(entry)
		roof["traffic"] = traffic
		# ONE line, under 2 KB (the driver keeps a 2,000-character tail of stdout): the headline, its roofline with every other
		# configuration as [value, kernel_ms, frac] under roofline.by_config, the CPU baseline.  The long form (workload texts,
		# phases, notes, traffic sources) goes to stderr and to VK_BENCH_FULL (a file) when set.
		out = {
			"metric": "sentence-alignments/sec at d=300, |q|=10, |s|<=64; 1/2/4/8 GPU + %HBM roofline",
			"value": round(entry["value"], 1), "unit": "sentence-alignments/sec",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": round(entry["ms_per_step"], 4),
			# the timed region is short (steps x ~3 ms): the median and the shortest / longest time between two completed steps say
			# how far a single slow step (or a slow box) moved `value`
			"ms_per_step_median": None if entry["ms_per_step_median"] is None else round(entry["ms_per_step_median"], 4),
			"step_gap_ms": None if entry["step_gap_ms"] is None else [round(x, 4) for x in entry["step_gap_ms"]],
			"higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
			"dtype": spec["prec"], "data": "synthetic",
			"config": {"workload": entry["workload"], "name": spec["name"],
				"parallelism": f"corpus shards x{world}, RCCL all-gather of k records"},
			"roofline": roof,
			"ranks": ranks,
		}
		if selfcheck is not None:
			out["selfcheck"] = selfcheck
		full = dict(out)
		full["config"] = dict(out["config"], sentences_per_gpu=n_sent, sentences_total=n_sent * world, len_s=[spec["min_len"], spec["max_len"]], len_t=spec.get("len_t", LEN_T), d=spec["d"], k=K_MATCHES,
			algorithm=spec["alg"], locality=spec["locality"], gap=spec["gap"], queries_per_step=max(1, int(spec.get("batch", 0))))
		full["roofline"] = dict(entry["roofline"], traffic=traffic, traffic_source=traffic_source)
		full["tokens_per_gpu"] = entry["tokens_per_gpu"]
		full["ms_per_step_median"] = entry["ms_per_step_median"]
		full["phases_ms_mean"] = entry["phases_ms_mean"]

	# ---- the other SURVEY 8(d) configurations, one GPU, each on its own resident corpus --------------------------
	if world == 1 and dist is None and not args.no_extra:
		configs = {}
		for key in [k for k in args.extra.split(",") if k and k in WORKLOADS and k != args.config]:
			w = WORKLOADS[key]
			n_x = max(w.get("min_n", 4096), int(w["n_sent"] * args.extra_scale))
			# a timed region of a few steps of 3.5 ms is at the mercy of whatever the previous configuration left running in the
			# driver (freed corpora: the static layout once showed 9 ms per step over 12 steps, 3.44 over 60): at least
			# --extra-min-ms of timed steps, a third of that as warm-up
			avg_len = (w["min_len"] + w["max_len"]) / 2
			if w.get("batch") and w["alg"] == "align":
				est_ms = -(-w["batch"] // w.get("per_pass", 2)) * n_x * avg_len * w["d"] * 2 / (0.6 * HBM_PEAK) * 1e3
			elif w.get("batch"):
				est_ms = 2.0 * n_x * avg_len * w["batch"] * LEN_T * w["d"] / (0.5 * MFMA_BF16_PEAK) * 1e3
			elif w.get("layout") == "static":
				est_ms = n_x * avg_len * LEN_T / 370e9 * 1e3
			elif w.get("len_t", LEN_T) > 32:
				est_ms = 9.0 * n_x / 1e6
			else:
				est_ms = n_x * avg_len * w["d"] * (4 if w["prec"] == "f32" else 2) / (w.get("rate_frac", 0.8) * HBM_PEAK) * 1e3
			x_steps = max(args.extra_steps, int(np.ceil(args.extra_min_ms / est_ms)))
			x_warmup = max(args.extra_warmup, int(np.ceil(args.extra_min_ms / 3 / est_ms)))
			try:
				_, e = measure(key, n_x, x_warmup, x_steps, None, keep=keep)
				if e["roofline"]["bound"] == "hbm" and w.get("layout") != "static":
					e["roofline"]["traffic"], e["roofline"]["traffic_source"] = traffic_of(w["name"], w["gap"], n_x, w["n_sent"])
			except Exception as ex:   # a configuration that fails is reported, the headline stands
				e = {"workload": describe(w, n_x), "error": f"{type(ex).__name__}: {ex}"}
			configs[w["name"]] = e
		full["configs"] = configs
		# [value, kernel_ms, frac of the roofline that bounds it, which roofline] per configuration; units as the headline's
		# (config4: pairs/s against the MFMA peak; "valu": VALU instructions/s against the issue peak, null without a counter pass)
		out["roofline"]["by_config"] = {name: ([round(e["value"], 1), round(e["kernel_ms"], 4), None if e["roofline"]["frac"] is None else round(e["roofline"]["frac"], 4),
			e["roofline"]["bound"]] if "error" not in e else "error") for name, e in configs.items()}
	if keep.get("shard"):
		keep["shard"][0].close()
		keep.clear()
		torch.cuda.empty_cache()

	if rank == 0:
		if not args.no_cpu_baseline and world == 1:
			cb = cpu_baseline(spec)
			full["cpu_baseline"] = cb
			out["cpu_baseline"] = {k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in cb.items() if k_ != "blas_note"}
		print("[bench full] " + json.dumps(full), file=sys.stderr)
		if os.environ.get("VK_BENCH_FULL"):
			with open(os.environ["VK_BENCH_FULL"], "w") as f:
				json.dump(full, f, indent=1)
		line = json.dumps(out, separators=(",", ":"))
		if len(line) > 1900:   # never let the tail cut the front of the line: drop the least needed parts first
			for drop in (("ranks", "visible"), ("cpu_baseline", "sample"), ("roofline", "kernel"), ("config", "parallelism")):
				out.get(drop[0], {}).pop(drop[1], None)
				line = json.dumps(out, separators=(",", ":"))
				if len(line) <= 1900:
					break
		print(line)
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
