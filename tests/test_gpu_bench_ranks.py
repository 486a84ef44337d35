"""-m gpu: bench.py's N > 1 code path under the test tier -- process-group init, stdout kept for the ONE JSON line, query broadcast,
batched exchange (shards.allgather_start / _finish), all_reduce(MAX) of the elapsed time, and the self-check of the sharded answer.
Two ranks over gloo on the one GPU of the box (VK_BENCH_BACKEND=gloo: the records travel through host memory; the RCCL calls themselves
are covered with one rank in test_gpu_shards_nccl.py), launched exactly as the driver launches N ranks: a fresh child process per run
(never an exec of this process)."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def run_bench(n, extra, env_extra=None, timeout=600):
	env = dict(os.environ, VK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
	env.update(env_extra or {})
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
		"--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra
	r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
	assert r.returncode == 0, r.stderr[-4000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, r.stdout[-2000:]           # ONE line on stdout, whatever the ranks and the backend print
	assert len(lines[0]) < 2000, len(lines[0])         # the driver keeps a 2,000-character tail
	return json.loads(lines[0])


@pytest.mark.parametrize("config,locality", [("2", None), ("3", None)])
def test_two_ranks_bench_line_and_selfcheck(config, locality):
	out = run_bench(2, ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--sentences", "65536", "--config", config, "--selfcheck"])
	assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
	assert out["value"] > 0 and out["ms_per_step"] > 0 and out["scaling"] == "weak"
	assert out["metric"].startswith("sentence-alignments/sec") and out["unit"] == "sentence-alignments/sec"
	assert out["ranks"]["world_seen_by_backend"] == 2 and out["ranks"]["launched"] == 2 and out["ranks"]["backend"] == "gloo"
	assert len(out["ranks"]["devices"]) == 2
	assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["bound"] == "hbm"
	# value = the pairs of BOTH ranks over the max-over-ranks time
	assert abs(out["value"] - 2 * 65536 * 6 / (out["ms_per_step"] * 6e-3)) / out["value"] < 1e-3
	# the merged result set of a fixed query == the answer of one corpus holding both shards (ids, scores, tracebacks: bit for bit)
	sc = out["selfcheck"]
	assert sc["ok"], sc
	assert sc["n"] == 10 and sc["merged"] == sc["one_corpus"]
	assert sc["ranks_with_winners"] == [0, 1], sc       # winners from both shards: the merge did merge


@pytest.mark.parametrize("config", ["3", "5"])
def test_two_ranks_strong_scaling(config):
	"""--scaling strong: the configuration's TOTAL (here --sentences 131072, standing in for config 3's 10 M / config 5's 4 M) is
	divided among the ranks; the line says "strong", value = the total's pairs over the max-over-ranks time, the self-check holds"""
	total = 131072 if config == "3" else 32768
	out = run_bench(2, ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--sentences", str(total), "--config", config, "--scaling", "strong", "--selfcheck"])
	assert out["n_gpus"] == 2 and out["scaling"] == "strong"
	assert abs(out["value"] - total * 6 / (out["ms_per_step"] * 6e-3)) / out["value"] < 1e-3    # (total // 2) sentences on each of 2 ranks
	assert out["ms_per_step_median"] > 0 and len(out["step_gap_ms"]) == 2 and out["step_gap_ms"][0] <= out["ms_per_step_median"] <= out["step_gap_ms"][1] + 1e-9
	sc = out["selfcheck"]
	assert sc["ok"], sc
	assert sc["merged"] == sc["one_corpus"] and set(sc["ranks_with_winners"]) <= {0, 1}, sc


def test_one_rank_over_rccl_runs_the_multi_rank_code_path():
	"""VK_BENCH_FORCE_DIST=1: the N > 1 code path of bench.py with the backend the driver's multi-GPU run uses (torch.distributed
	"nccl" = RCCL: communicator with a high-priority stream, stdout kept clean of RCCL's banner, all_gather_object of the ranks'
	devices, the batched exchange on the device, all_reduce(MAX) of the elapsed time), one rank on the one GPU"""
	env = dict(os.environ, VK_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
	for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
		env.pop(k, None)
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--sentences", "65536", "--selfcheck"],
		cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stderr[-4000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, r.stdout[-2000:]
	out = json.loads(lines[0])
	assert out["n_gpus"] == 1 and out["value"] > 0
	assert out["ranks"]["backend"] == "nccl" and out["ranks"]["world_seen_by_backend"] == 1 and out["ranks"]["devices"] == [0]
	assert out["selfcheck"]["ok"] and out["selfcheck"]["n"] == 10

