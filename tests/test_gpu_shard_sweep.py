"""-m gpu: the sharded index on the HIP backend, two and three ranks on one GPU (gloo between the processes), over the random
sessions of tests/test_gpu_index_sweep.py: Index.find and find_many of a rank's sharded index return exactly -- documents, slices,
scores and flows with `==` -- what the unsharded HIP index returns, on every rank.  VK_SWEEP_SCALE multiplies the seeds."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = int(os.environ.get("VK_SWEEP_SCALE", "1"))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_index_equals_unsharded_on_random_sessions(hip, tmp_path, world):
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	count = 25 * SCALE
	env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
		os.path.join(ROOT, "tests", "shard_sweep_worker.py"), str(tmp_path), str(1000 * world), str(count)]
	r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900 + 20 * count)
	assert r.returncode == 0, r.stderr[-4000:]
	for k in range(world):
		got = json.load(open(tmp_path / f"sweep_rank{k}.json"))
		assert got["fails"] == [], "\n".join(got["fails"][:5])
		assert got["done"] == count
