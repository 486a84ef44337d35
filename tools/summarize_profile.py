"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/.

python tools/summarize_profile.py gpurun_out/prof_r02 r02
  -> <dir>/summary/{tag}_pmc_score_kernel.csv, traffic.json (the kernel statistics are cut by the shell script)

traffic.json: per configuration ("<name>:<gap>", the keys bench.py looks up) the HBM bytes per launch of the scoring kernel:
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE counts half the bytes of a wide coalesced stream on gfx950
(MI355X_MICROARCH.md, HBM section) -- beside the algorithmic bytes bench.py states for the same launch.
"""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
	hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
	return hits[0] if hits else None


def bench_line(path):
	"""the JSON line bench.py printed in that pass"""
	if not os.path.exists(path):
		return None
	lines = open(path).read().splitlines()
	for line in reversed(lines):   # round 3: the ONE line on stdout is compact; the long form (config.gap, algorithmic bytes) goes to stderr
		if line.startswith("[bench full] {"):
			try:
				return json.loads(line[len("[bench full] "):])
			except ValueError:
				pass
	for line in reversed(lines):
		if line.startswith("{") and '"metric"' in line:
			try:
				return json.loads(line)
			except ValueError:
				pass
	return None


def main():
	src, tag = sys.argv[1], sys.argv[2]
	dst = os.path.join(src, "summary")
	os.makedirs(dst, exist_ok=True)
	traffic_all = {}
	rows_out = []
	for fetch_dir in sorted(glob.glob(os.path.join(src, "pmc_fetch_*"))):
		if not os.path.isdir(fetch_dir):
			continue
		c = os.path.basename(fetch_dir)[len("pmc_fetch_"):]
		line = bench_line(os.path.join(src, f"pmc_fetch_{c}.log"))
		if not line:
			continue
		per = {}
		for name, counter in ((f"pmc_fetch_{c}", "FETCH_SIZE"), (f"pmc_write_{c}", "WRITE_SIZE")):
			f = find(os.path.join(src, name), "counter_collection.csv")
			if not f:
				continue
			with open(f) as fh:
				for r in csv.DictReader(fh):
					if r.get("Counter_Name") != counter or "vk_score_kernel" not in r.get("Kernel_Name", ""):
						continue
					key = (r["Kernel_Name"], counter)
					per.setdefault(key, {}).setdefault(r["Dispatch_Id"], 0.0)
					per[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
		traffic = {}
		kernel = None
		for (kname, counter), d in per.items():
			vals = list(d.values())
			rows_out.append((c, kname, counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
			kernel = kname
			traffic[counter + "_KiB_avg"] = sum(vals) / len(vals)
			traffic["launches_measured"] = len(vals)
		if "FETCH_SIZE_KiB_avg" in traffic and "WRITE_SIZE_KiB_avg" in traffic:
			traffic["kernel"] = kernel
			traffic["hbm_bytes_per_launch"] = (2 * traffic["FETCH_SIZE_KiB_avg"] + traffic["WRITE_SIZE_KiB_avg"]) * 1024
			traffic["algorithmic_bytes_per_launch"] = line["roofline"].get("algorithmic_bytes_per_launch")
			traffic["ratio"] = traffic["hbm_bytes_per_launch"] / traffic["algorithmic_bytes_per_launch"]
			traffic["source"] = (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of bench.py --config {c} "
				f"--steps 4 --warmup 2 --no-extra, round {tag}")
			traffic["correction"] = "2 x FETCH_SIZE (gfx950 wide-stream under-count, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, x 1024"
			traffic_all[line["config"]["name"] + ":" + line["config"]["gap"]] = traffic
	with open(os.path.join(dst, f"{tag}_pmc_score_kernel.csv"), "w") as out:
		out.write("config,kernel,counter,launches,avg_KiB,min_KiB,max_KiB\n")
		for r in rows_out:
			out.write('%s,"%s",%s,%d,%.3f,%.3f,%.3f\n' % r)
	if traffic_all:
		with open(os.path.join(dst, "traffic.json"), "w") as out:
			json.dump(traffic_all, out, indent=1)
	print(json.dumps({k: round(v["ratio"], 4) for k, v in traffic_all.items()}))


if __name__ == "__main__":
	main()
