"""CPU tier: the scope guard of vk_query / vk_query_batch (vectorian_amd/csrc/vk_guard.h) -- no early return with copies in
flight.  The header is host-only; a g++ driver runs it against a fake stream, under AddressSanitizer (a drain that writes into a
buffer that is gone aborts).  Reference behaviour: exceptions unwind RAII state, vectorian/core/cpp/query.cpp:10-30."""

import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_guard_drains_the_stream_before_the_buffers_die(tmp_path):
	exe = str(tmp_path / "guard_driver")
	subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer",
		"-I", os.path.join(ROOT, "vectorian_amd", "csrc"), os.path.join(ROOT, "tests", "guard_driver.cpp"), "-o", exe], check=True)
	out = subprocess.run([exe], check=True, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
	lines = dict(l.split(" rc=", 1) for l in out.stdout.strip().split("\n"))
	# a failing body: the stream is drained, THEN the buffer dies
	assert lines["fail_with_copy_in_flight"] == "3: [drained] [buffer destroyed]"
	# success: the body synchronised itself, the guard does not touch the stream
	assert lines["success"] == "0: [drained] [buffer destroyed]"
	# an abort between passes (VK_ERR_ABORTED): drained before the three probes of its two buffers die
	assert lines["abort_between_passes"] == "6: [drained] [buffer destroyed] [buffer destroyed] [buffer destroyed]"
	# an exception: turned into a status, drained before the buffer dies
	assert lines["exception"] == "1: [exception: bad_alloc stand-in] [drained] [buffer destroyed]"
	assert lines["stable_addresses"].startswith("0:")


def test_entry_points_run_their_bodies_under_the_guard():
	"""vk_query and vk_query_batch are the guard's only callers: the bodies take the keep, no host vector that is handed to
	hipMemcpyAsync is a plain local any more"""
	import re
	csrc = os.path.join(ROOT, "vectorian_amd", "csrc")
	for name, body in (("vk_query.cpp", "query_body"), ("vk_batch.cpp", "query_batch_body")):
		text = open(os.path.join(csrc, name)).read()
		assert f"vk_run_guarded([&](vk_host_keep &keep) {{ return {body}(" in text
		# every host pointer of an asynchronous copy is a caller's array (out-> / q-> / qs[ / packed16), pinned library memory
		# (c->h_brows) or comes from the keep (a reference / pointer bound by keep.vec / keep.array)
		kept = set(re.findall(r"[&*](\w+) = \*?keep\.(?:vec|array)<", text))
		for m in re.finditer(r"hipMemcpy(?:2D)?Async\(([^;]*);", text):
			args = m.group(1)
			host = [a.strip() for a in args.split(",")]
			names = set(re.findall(r"\b([A-Za-z_]\w*)\b", host[0] + " " + host[1] + (" " + host[2] if "2D" in m.group(0) else "")))
			device_only = all(n.startswith("d_") or n in ("c", "size_t", "stride", "kk", "uint8_t", "st") for n in names)
			assert device_only or names & kept or names & {"out", "q", "qs", "q0", "packed16", "h_brows", "rows_dst", "src"}, m.group(0)
