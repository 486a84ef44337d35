#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one translation unit, from hipcc's resource remarks.
  python tools/kernel_regs.py vectorian_amd/csrc/vk_score_m0.hip [substring-filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

def main():
	args = sys.argv[1:]
	extra = []
	if "--" in args:
		i = args.index("--")
		args, extra = args[:i], args[i + 1:]
	src = args[0]
	flt = args[1] if len(args) > 1 else ""
	cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-pass-failed",
		"-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src] + extra
	err = subprocess.run(cmd, capture_output=True, text=True).stderr
	cur = None
	rows = {}
	for line in err.splitlines():
		m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
		if not m:
			if "error" in line:
				print(line)
			continue
		t = m.group(1)
		if t.startswith("Function Name:"):
			cur = t.split(":", 1)[1].strip()
			rows[cur] = {}
		elif cur and ":" in t:
			k, v = t.split(":", 1)
			rows[cur][k.strip()] = v.strip()
	for name, r in rows.items():
		dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
		if flt and flt not in dem:
			continue
		tot = int(r.get("VGPRs", 0)) + int(r.get("AGPRs", 0))
		print(f"{dem[:90]:90s} vgpr {r.get('VGPRs'):>4} agpr {r.get('AGPRs'):>3} alloc {(tot + 7) // 8 * 8:>4} sgpr {r.get('TotalSGPRs'):>4} scratch {r.get('ScratchSize [bytes/lane]'):>4} occ {r.get('Occupancy [waves/SIMD]')}")

if __name__ == "__main__":
	main()
