"""Worker of tests/test_shards_gloo.py: one rank = one corpus shard (oracle-backed test
double on CPU), gloo all-gather of the per-rank result sets, merged set written by rank 0."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(outdir):
	import torch.distributed as dist
	from fake_backend import OracleCorpus
	from vectorian_amd import core, shards, synth

	dist.init_process_group(backend="gloo")
	rank, world = dist.get_rank(), dist.get_world_size()
	corpus = synth.make_contextual_corpus(900, 2, 30, 800, 64)
	queries = synth.make_queries(corpus, 3, 6)
	off = corpus["sent_off"]
	a, b = shards.shard_ranges(900, world)[rank]
	X = corpus["X"][off[a]:off[b]]
	shard = OracleCorpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=64, n_tokens=len(X), n_sentences=b - a)
	shard.append_vectors(X, normalize=True)
	shard.set_sentences(off[a:b + 1] - off[a])
	shard.finalize()
	out = {}
	for qi, q in enumerate(queries):
		for name, loc in (("local", 0), ("global", 1)):
			top = shard.query(q["vectors"], locality=loc, gap_s=0.1, gap_t=0.1, max_matches=12,
				min_score=0.0 if loc == 0 else -100.0)
			if qi % 2 == 0:
				merged = shards.allgather_merge(top, a, 12)
			else:   # the pipelined form bench.py uses
				merged = shards.allgather_finish(shards.allgather_start(top, a, 12))
			out[f"{qi}_{name}_score"] = merged.score[:merged.n]
			out[f"{qi}_{name}_sentence"] = merged.sentence[:merged.n]
			out[f"{qi}_{name}_mapping"] = merged.mapping[:merged.n]
	# several result sets in one exchange (bench.py: one all-gather per few queries) == one exchange each
	tops = [shard.query(q["vectors"], locality=0, gap_s=0.1, gap_t=0.1, max_matches=12, min_score=0.0) for q in queries]
	for qi, merged in enumerate(shards.allgather_finish(shards.allgather_start(tops, a, 12))):
		one = shards.allgather_finish(shards.allgather_start(tops[qi], a, 12))
		assert merged.n == one.n and (merged.sentence[:one.n] == one.sentence[:one.n]).all() and (merged.score[:one.n] == one.score[:one.n]).all()
		out[f"{qi}_batched_sentence"] = merged.sentence[:merged.n]
		out[f"{qi}_batched_mapping"] = merged.mapping[:merged.n]
	np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
	dist.barrier()
	dist.destroy_process_group()


if __name__ == "__main__":
	main(sys.argv[1])
