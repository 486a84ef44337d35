"""Corpus shards across the GPUs of one node.

Sentences are independent given the query, so the corpus is split by contiguous sentence
ranges, one shard per process / GPU, read-only in that GPU's HBM.  The only exchange per
query is an all-gather of the k-record result sets (RCCL over xGMI when the process
group's backend is "nccl"; "gloo" on CPU for tests) followed by the same bounded merge the
reference performs on the host (ResultSet::extend, vectorian/core/cpp/result_set.h:70-93).
Global sentence index = shard offset + local index; every rank ends with the same set.
"""

import numpy as np

from vectorian_amd import core

def _layout(len_t):
	"""int32 words of one record: valid, score, raw, sentence(2), mapping i16[w] (w/2), edge_sim f32[w] (w), padded
	to a multiple of 4 words; w = query length rounded up to 16 (32 words for queries of at most 16 tokens)."""
	w = (max(1, len_t) + 15) // 16 * 16
	return w, (5 + w // 2 + w + 3) // 4 * 4


def shard_ranges(n_sentences, world):
	"""contiguous sentence ranges, sizes differing by at most one"""
	base, rem = divmod(n_sentences, world)
	out, a = [], 0
	for r in range(world):
		b = a + base + (1 if r < rem else 0)
		out.append((a, b))
		a = b
	return out


def shard_ranges_by_tokens(slice_len, world):
	"""contiguous slice ranges of (nearly) equal TOKEN counts (the work of a rank is the tokens it streams, so a
	corpus of ragged slices is cut where the running token count crosses r / world of the total, not at equal slice
	counts).  slice_len: tokens per slice; every rank computes the same cuts from the same array."""
	slice_len = np.asarray(slice_len, dtype=np.int64)
	n = len(slice_len)
	if n == 0:
		return [(0, 0)] * world
	cum = np.cumsum(slice_len)
	total = int(cum[-1])
	cuts = [0]
	for r in range(1, world):
		target = total * r / world
		i = int(np.searchsorted(cum, target, side="left"))      # first slice whose end reaches the target
		# the cut after slice i or before it, whichever leaves the running count closer to the target
		if i < n and (i == 0 or abs(int(cum[i]) - target) <= abs(int(cum[i - 1]) - target)):
			i += 1
		cuts.append(min(n, max(cuts[-1], i)))
	cuts.append(n)
	return [(cuts[r], cuts[r + 1]) for r in range(world)]


def pack_topk(top, sentence_offset, k):
	"""the k exchange records of a result set (native: vk_pack_records; the layout is _layout's)"""
	return core.pack_records(top, sentence_offset, k)


def pack_topk_numpy(top, sentence_offset, k):
	"""the record layout written out in numpy: the tests hold vk_pack_records against it"""
	w, words = _layout(top.len_t)
	buf = np.zeros((k, words), dtype=np.int32)
	n = top.n
	buf[:n, 0] = 1
	buf[:n, 1] = top.score[:n].view(np.int32)
	buf[:n, 2] = top.raw_score[:n].view(np.int32)
	buf[:n, 3:5] = (top.sentence[:n] + sentence_offset).astype(np.int64).view(np.int32).reshape(n, 2)
	m = np.full((n, w), -1, dtype=np.int16)
	m[:, :top.len_t] = top.mapping[:n]
	buf[:n, 5:5 + w // 2] = m.view(np.int32)
	e = np.zeros((n, w), dtype=np.float32)
	e[:, :top.len_t] = top.edge_sim[:n]
	buf[:n, 5 + w // 2:5 + w // 2 + w] = e.view(np.int32)
	return buf


def unpack_topk(buf, len_t):
	k = buf.shape[0]
	w, _ = _layout(len_t)
	t = core.TopK(k, len_t)
	n = int(buf[:, 0].sum())
	t.n = n
	b = np.ascontiguousarray(buf[:n])
	t.score[:n] = np.ascontiguousarray(b[:, 1]).view(np.float32)
	t.raw_score[:n] = np.ascontiguousarray(b[:, 2]).view(np.float32)
	t.sentence[:n] = np.ascontiguousarray(b[:, 3:5]).view(np.int64).reshape(n)
	t.mapping[:n] = np.ascontiguousarray(b[:, 5:5 + w // 2]).view(np.int16).reshape(n, w)[:, :len_t]
	t.edge_sim[:n] = np.ascontiguousarray(b[:, 5 + w // 2:5 + w // 2 + w]).view(np.float32).reshape(n, w)[:, :len_t]
	return t


FLAG_ABORTED = 1   # Query.abort() was seen by a rank: the query yields no matches on ANY rank (the ranks poll their flags at different times)


def allgather_start(tops, sentence_offset, k, group=None, device=None, flags=None):
	"""starts the exchange of this rank's result set(s) and returns a handle for allgather_finish; the collective runs
	while the caller scores the next queries.  `tops` may be a list: the result sets of several queries then travel
	in ONE all-gather (the records are tiny -- 1.3 KB per query at k = 10 -- and the exchange is latency-bound: per
	collective there is a launch, a kernel that has to find room beside the scoring kernel, and two small copies, so
	exchanging every few queries instead of every query divides that cost)."""
	import torch
	import torch.distributed as dist

	single = not isinstance(tops, (list, tuple))
	if single:
		tops = [tops]
	world = dist.get_world_size(group)
	if device is None:
		device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
	if len({_layout(t.len_t) for t in tops}) != 1:
		raise ValueError("the result sets of one exchange must have the same record size (query lengths in the same multiple of 16)")
	rows = np.empty((len(tops) * k, _layout(tops[0].len_t)[1]), dtype=np.int32)
	for i, t in enumerate(tops):
		core.pack_records(t, sentence_offset, k, out=rows[i * k:(i + 1) * k])
		# per-query flags of this rank ride in the spare last word of the query's first record (a record has three words of
		# padding behind the edge similarities); allgather_finish ORs them over the ranks into handle["flags_out"]
		rows[i * k, -1] = int(flags[i]) if flags is not None else 0
	on_gpu = torch.device(device).type == "cuda"
	if on_gpu:
		# pinned staging on both sides and a stream of its own: no pageable copies, nothing on the default stream
		stream = _exchange_stream(device)
		host = torch.empty(rows.shape, dtype=torch.int32, pin_memory=True)
		host.numpy()[:] = rows
		back = torch.empty((world * rows.shape[0], rows.shape[1]), dtype=torch.int32, pin_memory=True)
		with torch.cuda.stream(stream):
			send = host.to(device, non_blocking=True)
			recv = torch.empty((world * rows.shape[0], rows.shape[1]), dtype=torch.int32, device=device)   # concatenated along dim 0
			work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)
			work.wait()                          # orders the side stream behind the collective; does not block the host
			back.copy_(recv, non_blocking=True)
			done = torch.cuda.Event()
			done.record(stream)
		return dict(done=done, back=back, keep=(host, send, recv, work), world=world, k=k, n=len(tops), len_t=[t.len_t for t in tops], single=single)
	send = torch.from_numpy(rows)
	recv = torch.empty((world * rows.shape[0], rows.shape[1]), dtype=send.dtype)
	work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)
	return dict(work=work, back=recv, keep=(send,), world=world, k=k, n=len(tops), len_t=[t.len_t for t in tops], single=single)


_streams = {}


def _exchange_stream(device):
	import torch
	key = str(torch.device(device))
	if key not in _streams:
		_streams[key] = torch.cuda.Stream(device=device)
	return _streams[key]


def allgather_finish(handle):
	"""the merged global result set of every query of the exchange (one TopK, or a list if a list was started)"""
	if "done" in handle:
		handle["done"].synchronize()
	else:
		handle["work"].wait()
	world, k, n = handle["world"], handle["k"], handle["n"]
	allr = handle["back"].numpy().reshape(world, n, k, -1)
	handle["flags_out"] = [int(np.bitwise_or.reduce(allr[:, i, 0, -1])) for i in range(n)]
	# ResultSet.extend over every rank's records, one native call per query (vk_merge_records)
	out = [core.merge_records(np.ascontiguousarray(allr[:, i]), world, handle["len_t"][i], k) for i in range(n)]
	return out[0] if handle["single"] else out


def allgather_merge(top, sentence_offset, k, group=None, device=None):
	"""all ranks contribute their local result set; every rank returns the merged global one"""
	import torch
	import torch.distributed as dist

	world = dist.get_world_size(group)
	if device is None:
		device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
	send = torch.from_numpy(pack_topk(top, sentence_offset, k)).to(device)
	recv = torch.empty((world * k, send.shape[1]), dtype=send.dtype, device=device)   # concatenated along dim 0
	dist.all_gather_into_tensor(recv, send, group=group)
	allr = recv.cpu().numpy().reshape(world, k, send.shape[1])
	sets = [unpack_topk(allr[r], top.len_t) for r in range(world)]
	merged = core.merge_topk(sets, top.len_t, k)
	if getattr(top, "sim_rows", None) is not None:
		# transport winners (and the rows the debug hook asked for): their similarity rows [64 x W] and optimal plans [W x 64]
		# follow in a second all-gather (k x 2 x 64 W floats per rank: 80 KB at k = 10, W = 16), so that the merged winners
		# carry what the host states their SparseFlow / DenseFlow from (wmd.h:392-408, 228-248; wrd.h:120-135), whichever rank
		# scored them
		rows, w = top.sim_rows.shape[1], top.sim_rows.shape[2]
		pay = np.zeros((k, 2, rows * w), dtype=np.float32)
		pay[:top.n, 0] = top.sim_rows[:top.n].reshape(top.n, -1)
		pay[:top.n, 1] = top.plan[:top.n].reshape(top.n, -1)
		send2 = torch.from_numpy(pay).to(device)
		recv2 = torch.empty((world * k,) + pay.shape[1:], dtype=send2.dtype, device=device)
		dist.all_gather_into_tensor(recv2, send2, group=group)
		allp = recv2.cpu().numpy().reshape(world, k, 2, -1)
		where = {int(sets[r].sentence[j]): (r, j) for r in range(world) for j in range(sets[r].n)}
		merged.sim_rows = np.zeros((k, rows, w), dtype=np.float32)
		merged.plan = np.zeros((k, w, rows), dtype=np.float32)
		for i in range(merged.n):
			r, j = where[int(merged.sentence[i])]
			merged.sim_rows[i] = allp[r, j, 0].reshape(rows, w)
			merged.plan[i] = allp[r, j, 1].reshape(w, rows)
	return merged


def rows_allreduce(local_tops, merged_tops, sentence_offset, n_local, lens, group=None, device=None, with_plan=False):
	"""Second phase of a transport exchange: the similarity rows (and, exact transport, the optimal plans) of the MERGED winners,
	each contributed by the rank that scored it -- what the host states a winner's SparseFlow / DenseFlow from (wmd.h:392-408,
	228-248; wrd.h:120-135), so that flows read the same on every rank, whichever rank holds the slice.
	One all_reduce(SUM) over [queries x k x R x W] floats (x 2 with plans), zeros wherever a rank has nothing to say: k winners
	per query travel, not k per rank, and R is the longest MERGED winner, not the corpus's longest slice (`lens[i][j]`: tokens of
	merged winner j of query i -- every rank holds the slice table and computes the same).  (Round 2 all-gathered k x 2 x R_max x W
	per rank through pageable copies: 26 MB per rank and query at k = 100, R_max = 512.)
	Sets merged_tops[i].sim_rows [k x R x W], .plan [k x W x R] and .rows_room [k]."""
	import torch
	import torch.distributed as dist

	nq = len(merged_tops)
	if nq == 0:
		return
	if device is None:
		device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
	k = max(len(m.score) for m in merged_tops)
	R = max([int(np.max(l)) if len(l) else 0 for l in lens] + [1])
	W = max([t.sim_rows.shape[2] for t in local_tops if getattr(t, "sim_rows", None) is not None] + [16])
	# rows per winner a backend returned (batched calls: 64; vk_query: the longest slice of ITS shard): the ranks differ.  R = the
	# longest merged winner some rank can have rows for; with every winner travels the room its owner had (`rows_room`): a winner
	# longer than that has no rows anywhere and must read as one without -- not as rows cut short (the host would state a flow from them)
	cap = max([t.sim_rows.shape[1] for t in local_tops if getattr(t, "sim_rows", None) is not None and t.n > 0] + [0])
	Wt = torch.tensor([W, R, cap], dtype=torch.int64, device=device)
	dist.all_reduce(Wt, op=dist.ReduceOp.MAX, group=group)   # a rank without winners knows no W of its own
	W, R = int(Wt[0].item()), max(1, min(int(Wt[1].item()), max(1, int(Wt[2].item()))))
	planes = 2 if with_plan else 1
	flat = np.zeros(nq * k * planes * R * W + nq * k, dtype=np.float32)
	buf, room = flat[:nq * k * planes * R * W].reshape(nq, k, planes, R * W), flat[nq * k * planes * R * W:].reshape(nq, k)
	for i, (loc, mer) in enumerate(zip(local_tops, merged_tops)):
		if loc is None or getattr(loc, "sim_rows", None) is None or loc.n == 0:
			continue
		where = {int(sid): j for j, sid in enumerate(loc.sentence[:loc.n])}
		w_l = loc.sim_rows.shape[2]
		for j in range(mer.n):
			g = int(mer.sentence[j]) - sentence_offset
			if not (0 <= g < n_local) or g not in where:
				continue
			jl, ln = where[g], min(int(lens[i][j]), loc.sim_rows.shape[1], R)
			room[i, j] = float(loc.sim_rows.shape[1])
			rows = np.zeros((R, W), dtype=np.float32)
			rows[:ln, :w_l] = loc.sim_rows[jl][:ln]
			buf[i, j, 0] = rows.reshape(-1)
			if with_plan and getattr(loc, "plan", None) is not None:
				plan = np.zeros((W, R), dtype=np.float32)
				plan[:w_l, :ln] = loc.plan[jl][:, :ln]
				buf[i, j, 1] = plan.reshape(-1)
	t = torch.from_numpy(flat)
	on_gpu = torch.device(device).type == "cuda"
	if on_gpu:
		t = t.pin_memory().to(device, non_blocking=True)
	dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
	flat = t.cpu().numpy() if on_gpu else t.numpy()
	allb, room = flat[:nq * k * planes * R * W].reshape(nq, k, planes, R * W), flat[nq * k * planes * R * W:].reshape(nq, k)
	for i, mer in enumerate(merged_tops):
		mer.sim_rows = np.ascontiguousarray(allb[i, :, 0]).reshape(k, R, W)
		mer.plan = np.ascontiguousarray(allb[i, :, 1]).reshape(k, W, R) if with_plan else np.zeros((k, W, R), dtype=np.float32)
		mer.rows_room = np.minimum(room[i], R).astype(np.int64)   # slice tokens winner j has rows for (0: none)

