#!/usr/bin/env python3
"""bench.py -- sentence-alignments/sec of the brute-force alignment search on MI355X.

Workload (BASELINE.json configs[1], SURVEY 8d "Config 2"): per GPU 1,000,000 sentences x 32
tokens x 300-d bf16 (contextual layout, one vector per token occurrence), 10-token query,
local alignment, k = 10, min_score = 0.  A "step" is one query against the resident shard:
query upload -> fused similarity (MFMA) + DP kernel -> bounded result set -> flow of the
winners -> results on the host (N > 1: + all-gather of the per-rank result sets + merge).

  python bench.py --gpus N --steps K --warmup W [--gap exp5|linear] [--sentences n]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.  The corpus is sharded by sentence range (weak scaling: every
rank holds its own 1M-sentence shard); the only exchange is the all-gather of k records.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 300
LEN_S = 32
LEN_T = 10
VOCAB = 50000
K_MATCHES = 10
HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_PAIR = LEN_S * D * 2   # |s| * d * 2 bytes (bf16 token vectors, read once)


def gap_spec(name):
	if name == "linear":
		return 0.1, 0.1, "local alignment, linear gap u=0.1 (Smith-Waterman)"
	w = (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32)   # smooth_gap_cost(5)
	return ("table", w), ("table", w), "local alignment, general gap w(k)=1-2^(-k/5) (Waterman-Smith-Beyer)"


def build_shard(core, torch, n_sent, rank, device):
	"""synthetic shard generated on the GPU in chunks, handed to the library by device pointer"""
	from vectorian_amd import synth
	E = synth.make_vocab(VOCAB, D)                                   # seeded, shared by all ranks
	rng = np.random.default_rng(synth.SEED_CORPUS + 7919 * rank)
	n_tok = n_sent * LEN_S
	ids = synth.zipf_ids(n_tok, VOCAB, rng)
	off = np.arange(n_sent + 1, dtype=np.int64) * LEN_S
	corpus = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=D, n_tokens=n_tok, n_sentences=n_sent)
	E_dev = torch.from_numpy(E).to(device)
	gen = torch.Generator(device=device)
	gen.manual_seed(1000 + rank)
	chunk = 1 << 20
	for a in range(0, n_tok, chunk):
		b = min(a + chunk, n_tok)
		idx = torch.from_numpy(ids[a:b].astype(np.int64)).to(device)
		x = E_dev[idx] + 0.1 * torch.randn((b - a, D), device=device, generator=gen, dtype=torch.float32)
		x = x.contiguous()
		torch.cuda.synchronize()
		corpus.append_vectors_device(x.data_ptr(), b - a, core.VK_F32, normalize=True)
		del x, idx
	corpus.set_sentences(off)
	corpus.finalize()
	del E_dev
	torch.cuda.empty_cache()
	return corpus, E, ids


def make_queries(E, ids, n_queries, seed):
	rng = np.random.default_rng(seed)
	n_sent = len(ids) // LEN_S
	qs = []
	for i in range(n_queries):
		if i % 2 == 0:   # planted: noisy copy of 10 consecutive tokens of a corpus sentence
			s = int(rng.integers(0, n_sent))
			st = s * LEN_S + int(rng.integers(0, LEN_S - LEN_T + 1))
			qi = ids[st:st + LEN_T]
		else:
			qi = rng.integers(0, VOCAB, size=LEN_T)
		qs.append(np.ascontiguousarray(E[qi] + 0.05 * rng.standard_normal((LEN_T, D)).astype(np.float32), dtype=np.float32))
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


def cpu_baseline(gap_name, budget_s=12.0):
	"""The CPU restatement of the reference algorithm (oracle/, kind "port": the reference's
	own C++ path cannot be built offline) on a bounded sample of the same workload:
	all host cores, static sentence ranges per thread (the analogue of the reference's
	thread-per-document pool, vectorian/index.py:544-558), threads started once per batch."""
	from oracle import vk_oracle as vo
	from vectorian_amd import synth
	cores = usable_cores()
	n = 8192
	corpus = synth.make_contextual_corpus(n, LEN_S, LEN_S, VOCAB, D)
	Xb = synth.to_bf16_bits(synth.normalize_rows(corpus["X"]))
	qs = [synth.to_bf16_bits(synth.normalize_rows(q["vectors"])) for q in synth.make_queries(corpus, 16, LEN_T)]
	gs, gt, _ = gap_spec(gap_name)
	kw = dict(layout=vo.LAYOUT_CONTEXTUAL, d=D, sent_off=corpus["sent_off"], X=Xb, locality=vo.LOCAL,
		gap_s=gs, gap_t=gt, max_matches=K_MATCHES, min_score=0.0, n_threads=cores)
	t0 = time.perf_counter()
	vo.find_many(Qs=qs[:2], **kw)                      # calibrate the batch size to the budget
	per_q = (time.perf_counter() - t0) / 2
	batch = int(max(2, min(4096, budget_s / max(per_q, 1e-6))))
	Qs = [qs[i % len(qs)] for i in range(batch)]
	t0 = time.perf_counter()
	vo.find_many(Qs=Qs, **kw)
	el = time.perf_counter() - t0
	# beside it: one thread, and the reference's own static layout (token ids + per-query table [V x |q|])
	def rate(budget, n_sent, **over):
		k2 = dict(kw, **over)
		t1 = time.perf_counter()
		vo.find_many(Qs=qs[:1], **k2)
		b = int(max(1, min(1024, budget / max(time.perf_counter() - t1, 1e-6))))
		t1 = time.perf_counter()
		vo.find_many(Qs=[qs[i % len(qs)] for i in range(b)], **k2)
		return n_sent * b / (time.perf_counter() - t1)
	single = rate(3.0, n, n_threads=1)
	n_st = 16 * n      # the per-query table over the vocabulary is amortised over the slices, as in the full workload
	st = synth.make_static_corpus(n_st, LEN_S, LEN_S, VOCAB, D)
	Eb = synth.to_bf16_bits(synth.normalize_rows(st["E"]))
	static = rate(3.0, n_st, layout=vo.LAYOUT_STATIC, X=None, sent_off=st["sent_off"], tok_id=st["tok_id"], E=Eb)
	return {
		"value": n * batch / el, "unit": "sentence-alignments/sec", "cores": cores, "kind": "port",
		"single_thread_value": single, "static_layout_value": static,
		"sample": f"{batch} queries x {n} sentences x {LEN_S} tokens x {D}-d (same generator as the GPU workload), "
			f"{cores} threads, {el:.1f} s; CPU restatement of the reference algorithm (reference not runnable offline); "
			f"single_thread_value: the same on one thread; static_layout_value: token ids + per-query table over {n_st} slices, {cores} threads"}


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=40)
	ap.add_argument("--warmup", type=int, default=5)
	ap.add_argument("--gap", choices=["exp5", "linear"], default="exp5")
	ap.add_argument("--sentences", type=int, default=1000000, help="sentences per GPU")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-pipeline", action="store_true", help="one handle, one query at a time")
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

	n_sent = args.sentences
	corpus, E, ids = build_shard(core, torch, n_sent, rank, device)
	queries = make_queries(E, ids, args.steps + args.warmup, seed=3456)
	if dist is not None:   # one query stream for the whole job: rank 0's
		qt = torch.from_numpy(np.stack(queries)).to(xdev)
		dist.broadcast(qt, src=0)
		queries = list(qt.cpu().numpy())
	gs, gt, gap_desc = gap_spec(args.gap)

	from vectorian_amd import shards

	# Three handles on the resident shard (vk_corpus_view: shared arrays, own stream and workspaces), one host thread each:
	# query i + 1 is scored while the result set of query i is selected, retraced and copied out (measured: 331 M/s with
	# two handles, 342 M/s with three, 337 M/s with four).  Every query is
	# complete (top-k with flow on the host; with several ranks: merged across ranks) inside the timed region.
	from concurrent.futures import ThreadPoolExecutor
	n_handles = 1 if args.no_pipeline else max(1, int(os.environ.get("VK_BENCH_HANDLES", "3")))
	handles = [corpus] + [corpus.view() for _ in range(n_handles - 1)]
	pool = ThreadPoolExecutor(max_workers=len(handles))
	inflight = []      # futures of submitted queries, oldest first
	pending = []       # exchanges of earlier queries, in flight while later ones are scored (oldest first)
	GATHER_BATCH = max(1, int(os.environ.get("VK_BENCH_GATHER_BATCH", "4")))   # result sets of this many queries travel in one all-gather
	GATHER_DEPTH = int(os.environ.get("VK_BENCH_GATHER_DEPTH", "1"))   # a collective gets this many further exchanges to complete before anyone waits for it
	unsent = []        # result sets of finished queries, waiting for their exchange
	submitted = [0]
	score_ms = []

	def run_query(h, q):
		top = h.query(q, locality=core.Locality.LOCAL, gap_s=gs, gap_t=gt, q_normalize=True,
			max_matches=K_MATCHES, min_score=0.0, want_flow=True)
		return top, h.last_timings()["score_ms"]

	def exchange():
		if unsent:
			# per-rank result sets -> all ranks (RCCL all-gather over xGMI), then ResultSet.extend
			pending.append(shards.allgather_start(list(unsent), rank * n_sent, K_MATCHES, device=xdev))
			unsent.clear()

	def drain(keep=0):
		merged = None
		while len(pending) > keep:
			merged = shards.allgather_finish(pending.pop(0))[-1]   # global top-k of earlier queries on every rank
		return merged

	prof = {"wait": 0.0, "start": 0.0, "finish": 0.0} if os.environ.get("VK_BENCH_PROFILE") else None

	def retire():
		t_a = time.perf_counter()
		top, ms = inflight.pop(0).result()
		score_ms.append(ms)
		if dist is None:
			return top
		t_b = time.perf_counter()
		unsent.append(top)
		if len(unsent) >= GATHER_BATCH:
			exchange()
		t_c = time.perf_counter()
		merged = drain(keep=GATHER_DEPTH)
		if prof is not None:
			prof["wait"] += t_b - t_a; prof["start"] += t_c - t_b; prof["finish"] += time.perf_counter() - t_c
		return merged

	def step(q):
		h = handles[submitted[0] % len(handles)]
		submitted[0] += 1
		inflight.append(pool.submit(run_query, h, q))
		if len(inflight) >= len(handles):
			return retire()
		return None

	def sync():
		while inflight:
			retire()
		exchange()
		drain()
		torch.cuda.synchronize()
		if dist is not None:
			dist.barrier()
			torch.cuda.synchronize()

	for i in range(args.warmup):
		step(queries[i])
		if i < len(handles):
			sync()   # the first query of a handle has no predecessor to queue behind: one at a time, so that no two
			         # scoring kernels share the GPU (afterwards a handle's kernel waits for its peer's on the device)
	sync()
	score_ms.clear()
	t0 = time.perf_counter()
	for i in range(args.steps):
		step(queries[args.warmup + i])
	sync()
	elapsed = time.perf_counter() - t0
	timings = corpus.last_timings()
	if prof is not None and rank == 0:
		print("bench.py host profile (s, warmup included):", prof, file=sys.stderr)

	if dist is not None:
		t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t.item())

	if rank == 0:
		total_pairs = n_sent * world * args.steps
		value = total_pairs / elapsed
		kern_s = float(np.mean(score_ms)) * 1e-3
		achieved = n_sent * ALG_BYTES_PER_PAIR / kern_s
		traffic = None
		prof = os.path.join(ROOT, "profiles", "traffic.json")
		if os.path.exists(prof):
			try:
				traffic = json.load(open(prof)).get(args.gap, {}).get("hbm_bytes_per_launch") if n_sent == 1000000 else None
			except Exception:
				traffic = None
		out = {
			"metric": "sentence-alignments/sec at d=300, |q|=10, |s|<=64; 1/2/4/8 GPU + %HBM roofline",
			"value": value, "unit": "sentence-alignments/sec",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": elapsed / args.steps * 1e3,
			"higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": "bf16", "data": "synthetic",
			"config": {
				"workload": f"{LEN_T}-token query over {n_sent} x {LEN_S}-token synthetic sentences per GPU, "
					f"{D}-d bf16 per-token vectors (contextual layout), {gap_desc}, top-{K_MATCHES} with flow",
				"sentences_per_gpu": n_sent, "len_s": LEN_S, "len_t": LEN_T, "d": D, "k": K_MATCHES,
				"gap": args.gap, "parallelism": f"corpus shards x{world}, RCCL all-gather of k records"},
			"roofline": {
				"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
				"frac": achieved / HBM_PEAK, "traffic": traffic,
				"kernel": "vk_score_kernel", "kernel_ms": kern_s * 1e3,
				"algorithmic_bytes_per_launch": n_sent * ALG_BYTES_PER_PAIR},
			"phases_ms_last_step": timings,
		}
		if not args.no_cpu_baseline and world == 1:
			out["cpu_baseline"] = cpu_baseline(args.gap)
		print(json.dumps(out))
	pool.shutdown()
	for h in handles[1:]:   # views first, then the handle that owns the arrays
		h.close()
	corpus.close()
	if dist is not None:
		dist.barrier()
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
